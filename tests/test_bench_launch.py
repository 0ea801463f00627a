"""bench.py's launch contract (no GPU): ``python bench.py --gpus N`` with no WORLD_SIZE in
the environment re-executes itself under torch.distributed.run with N ranks (the driver's
command form), and under an explicit torchrun launch it reads RANK / WORLD_SIZE from the
environment.  The ranks here join a gloo group and measure nothing (ASR_BENCH_CPU_STUB)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, **env):
    e = dict(os.environ, **env)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    out = subprocess.run(cmd, env=e, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=240, stdin=subprocess.DEVNULL)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_gpus_flag_self_launches_n_ranks():
    line = _run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '4', '--warmup', '1'],
                ASR_BENCH_CPU_STUB='1')
    assert line == {'stub': True, 'n_gpus': 2, 'steps': 4, 'warmup': 1, 'gpus_arg': 2}


def test_launch_command_is_the_drivers_form():
    line = _run([sys.executable, 'bench.py', '--gpus', '8', '--steps', '5', '--warmup', '2'],
                ASR_BENCH_DRY_LAUNCH='1')
    cmd = line['launch']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-6:] == ['--gpus', '8', '--steps', '5', '--warmup', '2']
    assert os.path.basename(cmd[-7]) == 'bench.py'


def test_explicit_torchrun_form():
    line = _run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
                 '2', '--master-addr', '127.0.0.1', '--master-port', '29577', 'bench.py', '--gpus',
                 '2', '--steps', '3', '--warmup', '1'], ASR_BENCH_CPU_STUB='1')
    assert line['n_gpus'] == 2 and line['gpus_arg'] == 2 and line['steps'] == 3


# ---- the ONE stdout line stays small (a 21.7 KB line left BENCH_r05.parsed = null)

def _canned_detail(pad=1):
    """A DETAIL object with every key a full N = 1 run produces; `pad` multiplies the prose."""
    prose = 'achieved = executed MFMA flop/s over the summed durations of every launch; ' * (6 * pad)
    roof_g = {'kernel': 'gemm_hlx_kernel<4,4,2> (256x256 tile; operands packed once into split-fp16 '
                        'planes): all GEMM launches of the step', 'bound': 'mfma', 'achieved': 865.1,
              'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': 0.346, 'traffic': 1179016581.5,
              'algorithmic_fp32_tflops': 288.4, 'launches_per_step': 38.5, 'avg_launch_ms': 0.6852,
              'ms_per_step': 26.38, 'algorithmic_tflop_per_step': 7.87,
              'chip_owning': {'achieved': 1067.0, 'frac': 0.4268, 'algorithmic_fp32_tflops': 355.7,
                              'launches_per_step': 22.5, 'avg_launch_ms': 0.5809, 'ms_per_step': 13.07,
                              'algorithmic_tflop_per_step': 4.653},
              'shared': {'achieved': 725.0, 'frac': 0.29, 'algorithmic_fp32_tflops': 241.7,
                         'launches_per_step': 16.0, 'avg_launch_ms': 0.819, 'ms_per_step': 13.31,
                         'algorithmic_tflop_per_step': 3.217},
              'note': prose, 'dominant_of': {'roofline_gemm_step': 26.4, 'roofline_lstm_fwd': 9.5,
                                             'roofline_lstm_bwd': 15.2}}
    roof_l = {'kernel': 'lstm_bwd_kernel_c<2, false> (two-dimensional split) (persistent BPTT of one '
                        'BiLSTM layer, both directions)', 'bound': 'latency (hand-off)',
              'achieved': 264.1, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': 0.1056,
              'pipe': 'fp16 MFMA (3 per fp32 product)', 'algorithmic_fp32_tflops': 88.0,
              'traffic': 4110000000.0, 'algorithmic_bytes': 2620000000.0, 'hbm_GBps': 861.2,
              'avg_launch_ms': 3.04, 'steps_per_launch': 999.0, 'us_per_timestep': 3.043,
              'flops_per_launch': 2.68e11, 'note': prose,
              'geometry': {'compact_launches_per_step': 4.0, 'compact_us_per_timestep': 3.21,
                           'default_launches_per_step': 1.0, 'default_us_per_timestep': 2.21,
                           'note': prose}}
    sub = {'value': 17520.3, 'unit': 'audio-seconds/s', 'ms_per_step': 18.27, 'steps': 20, 'warmup': 5,
           'arithmetic': prose, 'workload': 'cfg2: brsmv1 5xBiLSTM(256), MFCC-39, 28-class CTC',
           'roofline': dict(roof_l), 'roofline_lstm_fwd': dict(roof_l), 'roofline_lstm_bwd': dict(roof_l),
           'roofline_gemm_step': dict(roof_g),
           'roofline_gate_gemm': {'kernel': 'gemm_hlx_kernel<4,4,2> 63936x4096x1024', 'achieved': 1100.0,
                                  'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': 0.44,
                                  'algorithmic_fp32_tflops': 366.0, 'avg_launch_ms': 1.43, 'pack_ms': 0.39}}
    return {
        'metric': 'audio-seconds/sec trained (MFCC+BiLSTM+CTC)', 'value': 15081.2,
        'unit': 'audio-seconds/s', 'n_gpus': 1, 'steps': 20, 'warmup': 5, 'ms_per_step': 42.436,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32 (products: split-f16 hi+lo MFMA)', 'data': 'synthetic', 'arithmetic': prose,
        'config': {'workload': 'cfg3: 5xBiLSTM(512), log-mel-80, 28-class CTC, batch 64 x 10 s @16 kHz',
                   'baseline_config': 'configs[2] minus its conv front-end (BASELINE.md 3 / north-star '
                                      'targets); as written -> as_written / cfg3_conv',
                   'global_batch': 64, 'utterance_seconds': 10.0, 'frames': 999, 'dropout': 0.2,
                   'optimizer': 'adam(clipnorm=400)', 'parallelism': 'dp1', 'params': 27639836},
        'roofline': dict(roof_g), 'roofline_gemm_step': dict(roof_g), 'roofline_lstm_fwd': dict(roof_l),
        'roofline_lstm_bwd': dict(roof_l), 'fallbacks': 0, 'ranks_seen_by_rccl': 1,
        'allreduce_model': {'bytes': 110559360, 'schedule': prose, 'ring_8gpu_ms': 2.513,
                            'direct_8gpu_ms': 0.359, 'ring_share_of_step': 0.0592},
        'roofline_gate_gemm': dict(sub['roofline_gate_gemm'], note=prose),
        'roofline_ctc': {'kernel': 'ctc_logsoftmax + ctc_alpha_beta + ctc_grad', 'bound': 'hbm',
                         'achieved': 42.1, 'peak': 8000.0, 'unit': 'GB/s', 'frac': 0.00526,
                         'avg_ms': 0.34, 'share_of_step': 0.008, 'algorithmic_bytes': 14321664.0,
                         'traffic': 38300000.0, 'note': prose},
        'roofline_conv': {'kernel': prose, 'bound': 'mfma', 'achieved': 774.0, 'peak': 2500.0,
                          'unit': 'TFLOP/s', 'frac': 0.31, 'ms_per_step': 3.81, 'note': prose},
        'cfg3_conv': dict(sub, value=24361.0, ms_per_step=26.27), 'as_written': dict(sub),
        'cfg2': dict(sub), 'cfg2_n128': dict(sub, value=38415.0, ms_per_step=33.32),
        'exact_fp32': dict(sub, value=5306.5, ms_per_step=120.607),
        'predict_latency': {'utterance_seconds': 10.0, 'topology': 'brsmv1 5xBiLSTM(256), MFCC-39',
                            'n1_kernel_ms': 5.896, 'real_time_factor': 1696.1},
        'eval_beam': {'utterances': 64, 'frames': 999, 'classes': 28, 'device_width_100_s': 0.0575,
                      'host_width_100_s': 0.0125, 'same_strings_width_100': True,
                      'device_width_400_s': 0.27, 'host_width_400_s': 0.049,
                      'same_strings_width_400': True, 'host_threads': 64, 'default_decoder': 'host',
                      'default_width_100_s': 0.0125, 'default_width_400_s': 0.049},
        'dataset_build': {'utterances': 2048, 'utterance_seconds': 10.0, 'chunk': 64,
                          'features': 'logfbank80', 'value': 43980.1, 'unit': 'audio-seconds/s',
                          'seconds': 0.466, 'format': 'h5', 'file_MB': 655.1},
        'cpu_baseline': {'value': 8.49, 'unit': 'audio-seconds/s', 'cores': 256, 'blas_threads': 8,
                         'kind': 'port', 'utterances': 16,
                         'calibration_s_per_24_frames': {'8': 0.61, '32': 0.8, '128': 1.9, '256': 3.3},
                         'sample': prose,
                         'frontend_1core': {'value': 161.0, 'unit': 'audio-seconds/s', 'cores': 1,
                                            'sample': prose},
                         'frontend_allcores': {'value': 3256.0, 'unit': 'audio-seconds/s', 'cores': 256,
                                               'sample': prose},
                         'cfg1_step': {'value': 21.3, 'unit': 'audio-seconds/s', 'cores': 256,
                                       'blas_threads': 8, 'sample': prose}},
    }


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_line_is_small_and_round_trips(tmp_path, monkeypatch):
    import io
    bench = _bench_module()
    for pad in (1, 40):                       # 40 x: every prose field several 10 KB long
        detail = _canned_detail(pad)
        assert len(json.dumps(detail)) > 20000
        monkeypatch.setenv('ASR_BENCH_DETAIL', str(tmp_path / 'bench_detail.json'))
        out, err = io.StringIO(), io.StringIO()
        text = bench.emit(detail, out, err)
        assert out.getvalue() == text + '\n' and '\n' not in text
        assert len(text.encode()) < bench.LINE_LIMIT < 8192
        line = json.loads(text)
        for k in bench.CONTRACT_KEYS + ('config', 'roofline', 'cpu_baseline'):
            assert k in line, k
        assert line['value'] == 15081.2 and line['ms_per_step'] == 42.436 and line['steps'] == 20
        assert line['config']['workload'].startswith('cfg3') and 'model' not in line['config']
        roof = line['roofline']
        for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms'):
            assert k in roof, k
        # frac = achieved / peak over ALL launches of the family; the chip-owning subset apart
        assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-3
        assert roof['frac'] == 0.346 and roof['chip_owning']['frac'] == 0.4268
        cb = line['cpu_baseline']
        for k in ('value', 'unit', 'cores', 'kind', 'sample', 'blas_threads', 'utterances'):
            assert k in cb, k
        for k in ('as_written', 'cfg2', 'cfg2_n128', 'exact_fp32'):
            assert set(line[k]) >= {'value', 'ms_per_step', 'frac'}, k
        assert line['as_written']['ms_per_step'] == 26.27
        # no string in the line is prose
        def strings(o):
            if isinstance(o, dict):
                for v in o.values():
                    yield from strings(v)
            elif isinstance(o, str):
                yield o
        assert max(len(x) for x in strings(line)) <= 150
        # the detail went to the file; stderr only NAMES it (one short line that is not JSON), so
        # a reader keeping an 8 KB tail of stdout + stderr together still holds the whole line
        assert json.loads((tmp_path / 'bench_detail.json').read_text())['roofline_ctc']['frac'] == 0.00526
        assert err.getvalue().startswith('bench detail: ') and len(err.getvalue()) < 400
        assert len(err.getvalue()) + len(text) < 8192


def test_line_survives_failed_companions():
    bench = _bench_module()
    detail = _canned_detail()
    for k in ('cfg3_conv', 'cfg2', 'cfg2_n128', 'exact_fp32', 'predict_latency', 'eval_beam',
              'dataset_build'):
        detail[k] = {'error': 'Traceback ' + 'x' * 3000}
    del detail['cpu_baseline']
    line = bench.compact_line(detail)
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    assert line['cfg2'] == {'error': 'Traceback ' + 'x' * 107 + '...'}
    assert line['value'] == 15081.2 and line['roofline']['frac'] == 0.346
