// Error reporting + device info for libasr_hip.so (host code).
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void asr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* asr_last_error(void) { return g_err; }
extern "C" int asr_version(void) { return ASR_HIP_ABI_VERSION; }

extern "C" int asr_device_info(int* num_cus, int* lds_bytes_per_cu, char* arch,
                               int arch_len) {
  int dev = 0;
  ASR_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  ASR_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
  if (num_cus) *num_cus = prop.multiProcessorCount;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return ASR_OK;
}

extern "C" int asr_stream_create_cu_mask(const uint32_t* mask, int words, asr_stream_t* stream_out) {
  ASR_CHECK_ARG(mask && words > 0 && stream_out, "stream_create_cu_mask: bad arguments");
  hipStream_t st = nullptr;
  ASR_CHECK_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask));
  *stream_out = (asr_stream_t)st;
  return ASR_OK;
}

extern "C" int asr_stream_destroy(asr_stream_t stream) {
  ASR_CHECK_HIP(hipStreamDestroy((hipStream_t)stream));
  return ASR_OK;
}
