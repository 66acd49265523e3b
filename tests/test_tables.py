"""Constant tables: product copies == oracle copies == SURVEY appendix C CRCs."""
import ctypes as C
import re
import os
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _parse_f32(path, name):
    src = open(path).read()
    m = re.search(r"static const float %s\[(\d+)\] = \{(.*?)\};" % name, src, re.S)
    vals = [float.fromhex(t.rstrip("f")) for t in re.findall(r"-?0x[0-9a-fp.+-]+f", m.group(2))]
    assert len(vals) == int(m.group(1))
    return np.array(vals, dtype=np.float32)


APPENDIX_C = {
    "cs": 0x40A1F1A8, "ca": 0x2B144553, "is_ratios": 0x0159E1E9, "imdct_win": 0x33A8CA30,
    "cos_n12": 0x9F16EF93, "cos_n36": 0xA72E18BD, "synth_dtbl": 0x39C3A499,
}
PRODUCT_NAMES = {"cs": "kAliasCs", "ca": "kAliasCa", "is_ratios": "kIsRatios", "imdct_win": "kImdctWin",
                 "cos_n12": "kCosN12", "cos_n36": "kCosN36", "synth_dtbl": "kSynthD"}


def test_literal_tables_crc():
    for key, want in APPENDIX_C.items():
        a = _parse_f32(os.path.join(ROOT, "oracle", "oracle_tables.h"), "ot_" + key)
        b = _parse_f32(os.path.join(ROOT, "pdmp3_amd", "csrc", "tables_data.h"), PRODUCT_NAMES[key])
        assert zlib.crc32(a.tobytes()) & 0xFFFFFFFF == want, key
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), key


def test_libm_tables(oracle, emul):
    pow43 = np.zeros(8207, np.float32)
    t1 = np.zeros(304, np.float32)
    t2 = np.zeros(312, np.float32)
    emul.emul_tables(pow43.ctypes.data_as(C.c_void_p), t1.ctypes.data_as(C.c_void_p), t2.ctypes.data_as(C.c_void_p))
    assert np.array_equal(pow43.view(np.uint32), oracle.pow43().view(np.uint32))
    # SURVEY appendix C CRCs of the libm-derived tables
    assert zlib.crc32(pow43.tobytes()) & 0xFFFFFFFF == 0x0EF4BB44
    assert zlib.crc32(oracle.nwin().tobytes()) & 0xFFFFFFFF == 0x7B4FDD6B
    assert zlib.crc32(t2.tobytes()) & 0xFFFFFFFF == 0x34D802C9       # 2^(k/4), k = -266..45
    assert zlib.crc32(t1[:37].tobytes()) & 0xFFFFFFFF == 0xE8D0EE70  # 2^(-n/2), n = 0..36
    assert t1[299] > 0 and t1[300] == 0                               # binary32 underflow point
    assert emul.emul_ldexp_forms_exact() == 1                        # device ldexp forms == libm pow, whole range
