"""The C-ABI library loads and exports every symbol include/bonito_hip.h declares (no GPU calls)."""
import os
import re

from conftest import ROOT
from bonito_amd import _lib


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "bonito_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bh_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    names = _declared_symbols()
    assert len(names) >= 15
    handle = _lib.lib()
    for n in names:
        assert hasattr(handle, n), "libbonito_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "bonito_amd/_lib.py has no ctypes signature for %s" % n
    assert sorted(_lib.SIGNATURES) == names


def test_abi_version_and_error_string():
    handle = _lib.lib()
    assert handle.bh_abi_version() == 1
    assert isinstance(_lib.last_error(), str)


def test_layer_struct_size_matches_header():
    import ctypes
    assert ctypes.sizeof(_lib.bh_layer_t) == _lib.lib().bh_sizeof_layer() == 168


def test_host_packers_run_without_gpu():
    import ctypes as C
    import numpy as np
    handle = _lib.lib()
    H = 32
    w = np.arange(4 * H * H, dtype=np.float32).reshape(4 * H, H) / 1024.0
    out = np.zeros(4 * H * H, np.uint16)
    assert handle.bh_lstm_pack_whh(w.ctypes.data_as(C.c_void_p), H, out.ctypes.data_as(C.c_void_p)) == 0
    halves = out.view(np.float16)
    # slice 1, gate 2, kstep 0, lane 5 (row 5, kgroup 0), j=3  -> W[2H + 16 + 5][3]
    pos = (((1 * 4 + 2) * 1 + 0) * 64 + 5) * 8 + 3
    assert halves[pos] == np.float16(w[2 * H + 16 + 5, 3])
    # lane 21 = row 5, kgroup 1 -> column 8 + j
    pos = (((1 * 4 + 2) * 1 + 0) * 64 + 21) * 8 + 3
    assert halves[pos] == np.float16(w[2 * H + 16 + 5, 11])
    n = handle.bh_conv1d_packed_halves(16, 20, 19)
    assert n == 32 * 320
    cw = np.random.default_rng(0).standard_normal((20, 16, 19)).astype(np.float32)
    pk = np.zeros(n, np.uint16)
    assert handle.bh_conv1d_pack(cw.ctypes.data_as(C.c_void_p), 16, 20, 19, pk.ctypes.data_as(C.c_void_p)) == 0
    pkh = pk.view(np.float16).reshape(32, 320)
    assert pkh[7, 3 * 16 + 5] == np.float16(cw[7, 5, 3])
    assert (pkh[20:] == 0).all() and (pkh[:, 304:] == 0).all()
    assert handle.bh_lstm_pack_whh(w.ctypes.data_as(C.c_void_p), 33, out.ctypes.data_as(C.c_void_p)) != 0
    assert "multiple of 32" in _lib.last_error()
