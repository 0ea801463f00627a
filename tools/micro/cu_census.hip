// Which compute units does a CU-masked stream (hipExtStreamCreateWithCUMask) use?  Launches a
// census kernel on streams with a few masks and prints, per mask, how many workgroups ran on each
// XCC and how many distinct (XCC, SE, CU) places were seen -- the bit -> place numbering is not
// documented, and engine.backward wants the recurrence and the GEMMs beside it on DIFFERENT XCDs
// (separate L2s).      hipcc --offload-arch=gfx950 -O2 tools/micro/cu_census.hip -o /tmp/cu_census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <set>
#include <vector>

__global__ void census(unsigned* out, int spin) {
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 0xf;   // XCC_ID[3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));          // HW_ID
    out[blockIdx.x] = (xcc << 16) | (hw & 0xffff);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  }
}

static void run(const char* name, const std::vector<uint32_t>& mask, unsigned* dev, int blocks) {
  hipStream_t st;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
    printf("%s: stream creation failed\n", name);
    return;
  }
  hipMemsetAsync(dev, 0xff, blocks * sizeof(unsigned), st);
  hipLaunchKernelGGL(census, dim3(blocks), dim3(64), 65536, st, dev, 3000);   // 64 KB LDS: <= 2 per CU
  hipStreamSynchronize(st);
  std::vector<unsigned> h(blocks);
  hipMemcpy(h.data(), dev, blocks * sizeof(unsigned), hipMemcpyDeviceToHost);
  int per_xcc[16] = {0};
  std::set<unsigned> places;
  int mod_ok = 0;
  for (int b = 0; b < blocks; ++b) {
    const unsigned xcc = h[b] >> 16, hw = h[b] & 0xffff;
    per_xcc[xcc & 15]++;
    places.insert((xcc << 16) | (hw & 0xff00));          // (xcc, se, sh, cu)
    if ((int)xcc == b % 8) ++mod_ok;
  }
  printf("%-28s blocks per XCC:", name);
  for (int x = 0; x < 8; ++x) printf(" %4d", per_xcc[x]);
  printf("  | distinct (xcc,se,cu) places %3zu | block b on XCC b%%8: %d of %d\n", places.size(), mod_ok, blocks);
  hipStreamDestroy(st);
}

int main() {
  unsigned* dev;
  hipMalloc(&dev, 4096 * sizeof(unsigned));
  const int words = 8;                                   // 256 CUs
  std::vector<uint32_t> m(words);
  auto fill = [&](auto pred) { for (int w = 0; w < words; ++w) { m[w] = 0; for (int i = 0; i < 32; ++i) if (pred(w * 32 + i)) m[w] |= 1u << i; } };
  fill([](int i) { return true; });            run("all 256", m, dev, 1024);
  fill([](int i) { return i < 128; });         run("bits 0..127", m, dev, 1024);
  fill([](int i) { return i >= 128; });        run("bits 128..255", m, dev, 1024);
  fill([](int i) { return i < 32; });          run("bits 0..31", m, dev, 1024);
  fill([](int i) { return (i % 8) < 4; });     run("i % 8 < 4", m, dev, 1024);
  fill([](int i) { return (i % 8) >= 4; });    run("i % 8 >= 4", m, dev, 1024);
  fill([](int i) { return (i % 8) == 0; });    run("i % 8 == 0", m, dev, 1024);
  fill([](int i) { return ((i / 8) % 2) == 0; }); run("even groups of 8", m, dev, 1024);
  fill([](int i) { return ((i / 32) % 2) == 0; }); run("even groups of 32", m, dev, 1024);
  return 0;
}
