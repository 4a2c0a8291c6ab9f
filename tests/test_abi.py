"""The C-ABI library loads and exports every symbol include/bonito_hip.h declares (no GPU calls)."""
import os
import pytest
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


def test_generated_gemm_stream_matches_its_generator(tmp_path):
    """bonito_amd/csrc/gemm_ktile_mfma.inc (one K-tile of gemm_w4_kernel as one instruction stream, with and without the previous
    tile's parked stores) is what tools/gen_gemmstep.py writes with its defaults."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "gemm_ktile_mfma.inc"
    subprocess.run([sys.executable, os.path.join(root, "tools", "gen_gemmstep.py"), "--out", str(out)], check=True, capture_output=True)
    committed = open(os.path.join(root, "bonito_amd", "csrc", "gemm_ktile_mfma.inc")).read()
    assert out.read_text() == committed
    # 64 MFMAs, 32 fragment reads, 16 DMA pieces and exactly one barrier per K-tile function
    body = committed.split("__device__ __forceinline__ void gemm_ktile(")[1].split("__device__ __forceinline__ void")[0]
    assert body.count("v_mfma_f32_32x32x16_f16") == 64 and body.count("ds_read_b128") == 32
    assert body.count("global_load_lds_dwordx4") == 16 and body.count("s_barrier") == 1
    # the same K-tile on 16x16x32 MFMAs (--tile16): 128 MFMAs, the same reads / pieces / barrier; every function one asm statement
    out16 = tmp_path / "gemm_ktile16_mfma.inc"
    subprocess.run([sys.executable, os.path.join(root, "tools", "gen_gemmstep.py"), "--tile16", "--out", str(out16)], check=True, capture_output=True)
    committed = open(os.path.join(root, "bonito_amd", "csrc", "gemm_ktile16_mfma.inc")).read()
    assert out16.read_text() == committed
    body = committed.split("__device__ __forceinline__ void gemm_ktile16(")[1].split("__device__ __forceinline__ void")[0]
    assert body.count("v_mfma_f32_16x16x32_f16") == 128 and body.count("ds_read_b128") == 32
    assert body.count("global_load_lds_dwordx4") == 16 and body.count("s_barrier") == 1 and body.count("asm volatile(") == 1
    # nothing of K-tile g is read behind the barrier, nothing of K-tile g + 1 in front of it
    front, behind = body.split("s_barrier")
    assert front.count("ds_read_b128") == 22 and behind.count("ds_read_b128") == 10
    assert front.index("%[san]") > front.rindex("%[ra0] offset") and "%[san]" not in behind       # the next K-tile's address replaces ra0 behind its last use


def test_tile16_gemm_stream_waits_for_every_fragment_it_uses():
    """An independent reading of the generated 16x16x32 K-tile streams (all variants in gemm_ktile16_mfma.inc): replay the instruction
    text with the LDS return queue (reads return in order; `s_waitcnt lgkmcnt(n)` = at most n outstanding) and check that (i) no MFMA
    reads a fragment register whose ds_read may still be in flight, (ii) no ds_read overwrites a register while an OLDER read into the
    same register is still in flight, (iii) every accumulator is used exactly once per 32-deep slab, by the W tile / X tile pair its name
    says, (iv) the barrier is preceded by lgkmcnt(0) and vmcnt(8), and nothing of the next K-tile's stage (addresses made from san / sbn)
    is read in front of it."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "bonito_amd", "csrc", "gemm_ktile16_mfma.inc")).read()
    bodies = re.findall(r"asm volatile\(\"(.*?)\"\n\s*: ", text, flags=re.S)
    assert len(bodies) == 8                      # plain, first, and the three parked-store flavours of each
    for body in bodies:
        ins = [ln.strip().strip('"').replace("\\n\\t", "").strip() for ln in body.splitlines()]
        ins = [i for i in ins if i]
        fifo, used, next_stage_addr = [], {}, set()
        seen_barrier = False
        for k, i in enumerate(ins):
            m = re.match(r"v_mfma_f32_16x16x32_f16 %\[(c\d_\d)\], %\[(fb(\d)_(\d))\], %\[(fa(\d))\], (?:%\[(c\d_\d)\]|0)$", i)
            if m:
                c, fb, slab, j, fa, ti = m.group(1), m.group(2), int(m.group(3)), int(m.group(4)), m.group(5), int(m.group(6))
                assert fb not in fifo and fa not in fifo, (k, i, fifo)
                assert c == "c%d_%d" % (ti, j) and (m.group(7) in (None, c))
                used.setdefault(c, []).append(slab)
                continue
            m = re.match(r"ds_read_b128 %\[(\w+)\], %\[(\w+)\] offset:\d+$", i)
            if m:
                assert m.group(1) not in fifo, (k, i)
                if m.group(2) in next_stage_addr:
                    assert seen_barrier, (k, i)
                fifo.append(m.group(1))
                continue
            m = re.match(r"s_waitcnt lgkmcnt\((\d+)\)$", i)
            if m:
                del fifo[:max(0, len(fifo) - int(m.group(1)))]
                continue
            m = re.match(r"v_add_u32_e32 %\[(\w+)\], %\[(san|sbn)\], ", i)
            if m:
                next_stage_addr.add(m.group(1))
                continue
            m = re.match(r"v_add_u32_e32 %\[(\w+)\], %\[(sa|sb)\], ", i)
            if m:
                next_stage_addr.discard(m.group(1))
                continue
            if i == "s_barrier":
                assert not fifo and ins[k - 1] == "s_waitcnt vmcnt(8)" and ins[k - 2] == "s_waitcnt lgkmcnt(0)"
                seen_barrier = True
        assert seen_barrier and not fifo                              # the stream ends with lgkmcnt(0)
        assert len(used) == 64 and all(v == [0, 1] for v in used.values())


@pytest.mark.parametrize("preset,name", [("plain", "ringstep3_mfma.inc"), ("paired", "ringstep3p_mfma.inc"), ("unrolled", "ringstep3u_mfma.inc"),
                                         ("rec", "ringstep3r_mfma.inc")])
def test_ring_step_streams_match_their_generator(tmp_path, preset, name):
    """bonito_amd/csrc/ringstep3_mfma.inc / ringstep3p_mfma.inc (the hand-scheduled instruction stream of a ring step of the paired
    recurrent kernel; the second one carries the section's DMA and validation too) are generated by tools/gen_ringstep.py
    --preset plain|paired: the committed files must be what the committed generator writes, and each must be ONE asm statement
    (the compiler can place nothing inside; tools/audit_ringstep.py is the guard for multi-statement streams)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_ringstep", os.path.join(root, "tools", "gen_ringstep.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    out = str(tmp_path / name)
    gen.main(["--preset", preset, "--out", out])
    committed = open(os.path.join(root, "bonito_amd", "csrc", name)).read()
    assert open(out).read() == committed
    assert committed.count("asm volatile(") == 1 and committed.count("v_mfma_f32_16x16x32_f16") == (36 if preset == "rec" else 72)
    # hazards the generator is responsible for: no transcendental result consumed by the next instruction
    lines = [ln.strip().strip('"').replace("\\n\\t", "") for ln in committed.splitlines() if ln.strip().startswith('"') or "asm volatile" in ln]
    for a, b in zip(lines, lines[1:]):
        if a.startswith(("v_exp_f32", "v_rcp_f32")) or "asm volatile(\"v_exp" in a:
            dst = a.split()[1].rstrip(",")
            assert dst not in b.split("//")[0].replace(",", " ").split()[2:], (a, b)


def test_unrolled_ring_step_stream_computes_what_the_paired_one_computes():
    """ringstep3u_mfma.inc (main loop of the paired kernel, unrolled over four steps) is scheduled differently from ringstep3p_mfma.inc
    (publish woven into the last MFMAs, plain stores, polls later, tile offsets as template constants) but must compute the same
    bits: every accumulator sees the same MFMAs in the same order, every cell the same gate arithmetic in the same order."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def body(name):
        text = open(os.path.join(root, "bonito_amd", "csrc", name)).read()
        text = text[text.index("asm volatile(") + len("asm volatile("):text.index("\n        : ")]
        return [ln.strip().strip('"').replace("\\n\\t", "") for ln in text.splitlines()]

    paired, unrolled = body("ringstep3p_mfma.inc"), body("ringstep3u_mfma.inc")
    for acc in ["v[232:235]", "v[236:239]", "v[240:243]", "%[xa0]", "%[xa1]", "%[xa2]"]:
        seq = [[ln for ln in b if ln.startswith("v_mfma") and ln.split()[1] == acc + ","] for b in (paired, unrolled)]
        assert len(seq[0]) == 12 and seq[0] == seq[1], acc
    for m in range(3):
        regs = re.compile(r"\bv(%s)\b|%%\[[ec]%d\]" % ("|".join(str(232 + 4 * m + i) for i in range(4)), m))
        seq = [[ln for ln in b if not ln.startswith(("v_mfma", "v_cvt_f16")) and regs.search(ln)] for b in (paired, unrolled)]
        assert len(seq[0]) == 35 and seq[0] == seq[1], m
    # the publish: the same three conversions, 2-byte writes and the 8-byte read; the stores plain in the unrolled stream
    for pat in ("v_cvt_f16_f32", "ds_write_b16", "ds_read_b64"):
        assert [ln for ln in paired if ln.startswith(pat)] == [ln for ln in unrolled if ln.startswith(pat)]
    assert sum(ln.startswith("global_store_dwordx2") for ln in unrolled) == 3 and not any("sc1" in ln for ln in unrolled if ln.startswith("global_store"))
    assert sum(ln.startswith("global_load_lds_dwordx4") for ln in unrolled) == 6
    # the validation's vmcnt leaves exactly the vector-memory operations issued behind the last poll outstanding (in-order counter), and the
    # read-back of the polled quarter comes behind it; the same rule in the generic stream
    for b in (paired, unrolled):
        polls = [i for i, ln in enumerate(b) if ln.startswith("global_load_lds_dwordx4") and "%[exo]" in ln]
        waits = [i for i, ln in enumerate(b) if ln.startswith("s_waitcnt vmcnt(")]
        assert len(polls) == 3 and len(waits) == 1 and waits[0] > polls[-1]
        younger = sum(ln.startswith(("global_load_lds", "global_store")) for ln in b[polls[-1] + 1:waits[0]])
        assert b[waits[0]] == "s_waitcnt vmcnt(%d)" % younger
        reads = [i for i, ln in enumerate(b) if ln.startswith("ds_read_b128 v[22")]
        assert len(reads) == 3 and min(reads) > waits[0]


def test_bs2_constants_header_matches_its_generator(tmp_path):
    """include/bh_bs2.h (polynomial coefficients of the deterministic exp / log shared by beam.hip and oracle/crf_oracle.c) is what
    tools/gen_bs2_coeffs.py writes; both sides include the SAME file, which is what makes the guide rows bit-identical."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "bh_bs2.h"
    subprocess.run([sys.executable, os.path.join(root, "tools", "gen_bs2_coeffs.py"), str(out)], check=True, capture_output=True)
    assert out.read_text() == open(os.path.join(root, "include", "bh_bs2.h")).read()
    for path in (os.path.join(root, "oracle", "crf_oracle.c"), os.path.join(root, "bonito_amd", "csrc", "beam.hip")):
        assert "bh_bs2.h" in open(path).read()


def test_profile_classes_of_the_binding_match_the_header():
    """bh_encoder_profile_read fills BH_PROF_CLASSES floats: the Python handle's class names (round 6: fc1 and the attention kernel have
    classes of their own) must be exactly the header's enumerators, in order."""
    import re
    from bonito_amd.engine import HipEncoder
    text = open(os.path.join(ROOT, "include", "bonito_hip.h")).read()
    enum = text[text.index("enum { BH_PROF_CONV"):]
    enum = enum[:enum.index("};")]
    enum = re.sub(r"/\*.*?\*/", "", enum, flags=re.S)
    names = re.findall(r"BH_PROF_(\w+)\s*=\s*(\d+)", enum)
    table = {name.lower(): int(v) for name, v in names}
    n = table.pop("classes")
    assert n == len(HipEncoder.PROF_CLASSES) == len(table)
    for i, cls in enumerate(HipEncoder.PROF_CLASSES):
        assert table[cls] == i, (cls, i, table)
