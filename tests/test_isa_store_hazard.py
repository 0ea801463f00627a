"""CPU: the gfx950 ISA hipcc generates for the kernels with 16-byte BUFFER stores must not hold
the store-data hazard the compiler leaves unpadded (a register in the soffset field and a VALU
write of the data registers within two instructions: tools/store_hazard_scan.py, DESIGN.md 5 --
measured on the GPU as stale store data in about one wave per launch).  Cross-compiles the two
recurrent-kernel sources to assembly (no GPU needed, ~50 s)."""
import shutil

import pytest

from tools import store_hazard_scan as S

RISKY = '''
_Zkern:                                 ; @_Zkern
\tv_pk_fma_f32 v[6:7], v[70:71], v[10:11], v[6:7]
\tbuffer_store_dwordx4 v[6:9], v53, s[52:55], s10 offen
\tv_pk_mul_f32 v[6:7], v[96:97], v[82:83] op_sel_hi:[0,1]
\ts_endpgm
'''
PADDED = RISKY.replace('s10 offen\n', 's10 offen\n\ts_nop 1\n')
NO_REG = RISKY.replace('s10 offen', '0 offen')           # (hipcc pads this form itself)
OTHER_REGS = RISKY.replace('v_pk_mul_f32 v[6:7]', 'v_pk_mul_f32 v[12:13]')


def test_scanner_recognises_the_pattern():
    assert S.scan(RISKY)[0][2] and S.scan(RISKY)[0][1] == 1
    assert S.scan(PADDED)[0][2] == []
    assert S.scan(NO_REG) == []
    assert S.scan(OTHER_REGS)[0][2] == []


@pytest.mark.timeout(600)
def test_recurrent_kernels_hold_no_unpadded_store_data_hazard():
    if shutil.which('/opt/rocm/bin/hipcc') is None:
        pytest.skip('no hipcc')
    srcs = S.sources()
    assert 'lstm_bwd.hip' in srcs
    seen = 0
    for src in srcs:
        for kern, n, risky in S.scan(S.isa(src)):
            seen += n
            assert not risky, (src, kern, risky)
    assert seen > 50        # the BPTT kernels' publishes were found at all
