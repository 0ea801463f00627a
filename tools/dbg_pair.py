import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_study_amd import ops
T, N, H = int(os.environ.get('T', 9)), 32, 256
rs = np.random.RandomState(0)
n_pad = 32
dev = 'cuda:0'
zx = torch.from_numpy(rs.randn(T, n_pad, 2, 4 * H).astype(np.float32)).to(dev)
U = torch.from_numpy((rs.randn(2, H, 4 * H) / np.sqrt(H)).astype(np.float32)).to(dev)
def run():
    y = torch.zeros(T, n_pad, 2 * H, device=dev)
    cell = torch.zeros(T, n_pad, 2, H, device=dev)
    gates = torch.zeros(T, n_pad, 2, 4 * H, device=dev)
    ws = ops.lstm_seq_fwd(zx, U, y, cell, gates, T, n_pad, H)
    ops.lstm_status(ws)
    return y.cpu().numpy().reshape(T, n_pad, 2, H)
os.environ['ASR_LSTM_PAIR'] = '0'
want = run()
for place in (2, 1):
    os.environ['ASR_LSTM_PAIR'] = '1'
    os.environ['ASR_LSTM_PAIR_PLACE'] = str(place)
    got = run()
    bad = got != want
    print('place', place, 'mismatches', bad.sum(), 'of', bad.size)
    if bad.any():
        for d in range(2):
            per_t = bad[:, :, d].reshape(T, -1).sum(1)
            print(' dir', d, 'per frame:', per_t.tolist())
            per_tile = [int(bad[:, 16 * b:16 * b + 16, d].sum()) for b in range(2)]
            print(' dir', d, 'per tile:', per_tile)
            bu = bad[:, :, d].sum((0, 1))
            print(' dir', d, 'units bad:', np.nonzero(bu)[0][:20].tolist(), '... count', int((bu > 0).sum()))
        idx = np.argwhere(bad)[0]
        print(' first', idx.tolist(), got[tuple(idx)], want[tuple(idx)], 'maxabs', np.abs(got - want).max())
