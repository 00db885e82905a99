"""CPU-only: the C-ABI library loads, exports every symbol include/opv_demod.h declares,
fails loudly without a GPU (no CPU fallback), and its host-side transmit chain is
bit-identical to the reference modulator (sha256 pins made by the reference binary)."""
import ctypes as C
import hashlib
import re
from pathlib import Path

import numpy as np
import pytest

from amd_lib import ROOT, load


@pytest.fixture(scope="module")
def amd():
    m = load()
    m.build()
    return m


def declared_symbols():
    text = (ROOT / "include" / "opv_demod.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(opv_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(amd):
    L = amd.lib()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/opv_demod.h but not exported"
    assert sorted(amd.EXPORTS) == names, "opv_amd.EXPORTS out of sync with the header"
    assert L.opv_abi_version() == 3


def test_struct_layouts_match_header(amd, tmp_path):
    """sizes and field offsets of the ctypes / numpy mirrors against what gcc makes of include/opv_demod.h"""
    import subprocess
    fields = {"opv_cfg": [f[0] for f in amd.Cfg._fields_], "opv_stream_state": [f[0] for f in amd.StreamState._fields_],
              "opv_frame_meta": list(amd.META_DTYPE.names), "opv_event": list(amd.EVENT_DTYPE.names)}
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT / "include" / "opv_demod.h"}"', "int main(void){"]
    for st, fs in fields.items():
        src.append(f'printf("{st} %zu\\n", sizeof({st}));')
        src += [f'printf("{st}.{f} %zu\\n", offsetof({st}, {f}));' for f in fs]
    src.append("return 0;}")
    (tmp_path / "l.c").write_text("\n".join(src))
    subprocess.run(["gcc", "-o", str(tmp_path / "l"), str(tmp_path / "l.c")], check=True)
    got = dict(ln.split() for ln in subprocess.run([str(tmp_path / "l")], capture_output=True, text=True, check=True).stdout.splitlines())
    assert int(got["opv_cfg"]) == C.sizeof(amd.Cfg) and int(got["opv_stream_state"]) == C.sizeof(amd.StreamState)
    assert int(got["opv_frame_meta"]) == amd.META_DTYPE.itemsize == 32 and int(got["opv_event"]) == amd.EVENT_DTYPE.itemsize == 32
    for f in fields["opv_cfg"]:
        assert int(got[f"opv_cfg.{f}"]) == getattr(amd.Cfg, f).offset, f
    for f in fields["opv_stream_state"]:
        assert int(got[f"opv_stream_state.{f}"]) == getattr(amd.StreamState, f).offset, f
    for f in fields["opv_frame_meta"]:
        assert int(got[f"opv_frame_meta.{f}"]) == amd.META_DTYPE.fields[f][1], f
    for f in fields["opv_event"]:
        assert int(got[f"opv_event.{f}"]) == amd.EVENT_DTYPE.fields[f][1], f


def test_no_gpu_means_loud_failure_not_fallback(amd):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(amd.OpvError) as e:
        amd.Demod(1, max_samples=1000)
    assert "-2" in str(e.value) or "HIP device" in str(e.value)   # OPV_ENODEV


def test_product_never_references_the_oracle():
    pkg = ROOT / "opv-cxx-demod_amd"
    for p in list(pkg.rglob("*.hip")) + list(pkg.rglob("*.cpp")) + list(pkg.rglob("*.h")) + list(pkg.rglob("*.py")) \
            + [pkg / "Makefile", ROOT / "include" / "opv_demod.h"]:
        t = p.read_text()
        assert "oracle/" not in t and "opv_oracle" not in t and "oro_" not in t, p


def test_host_tx_is_bit_identical_to_opv_mod(amd, golden, oracle):
    _, meta = golden
    pins = meta["opv_mod_bert_W5NYV"]
    for n in (10, 100):
        iq = amd.modulate(amd.bert_frames(n))
        assert iq.nbytes == pins[str(n)]["bytes"]
        assert hashlib.sha256(iq.tobytes()).hexdigest() == pins[str(n)]["sha256"]
    # and agrees with the oracle's independent generator on arbitrary payloads / callsigns
    rng = np.random.default_rng(5)
    fr = rng.integers(0, 256, (7, 134), dtype=np.uint8)
    assert np.array_equal(amd.modulate(fr), oracle.modulate(fr))
    for cs in ("W5NYV", "KB5MU-7", "a/b.c", "TOOLONGCALLSIGN", ""):
        assert np.array_equal(amd.bert_frames(2, cs, 0x123456, 9), oracle.bert_frames(2, cs, 0x123456, 9))


def test_raw_mode_kat_through_product_tx(amd, golden):
    arrays, meta = golden
    iq = amd.modulate(arrays["raw_kat_frames"])
    assert hashlib.sha256(iq.tobytes()).hexdigest() == meta["raw_kat"]["iq_sha256"]


def test_cli_binaries_built_and_usage(amd):
    import subprocess
    b = ROOT / "opv-cxx-demod_amd" / "bin"
    assert (b / "opv-demod").exists() and (b / "opv-mod").exists()
    r = subprocess.run([str(b / "opv-demod"), "-h"], capture_output=True)
    assert r.returncode == 0 and b"-s" in r.stderr and b"-o <hz>" in r.stderr
    out = subprocess.run([str(b / "opv-mod"), "-S", "W5NYV", "-B", "1"], capture_output=True).stdout
    assert len(out) == (2168 * 40 + 4000) * 4


def test_bench_starts_its_own_ranks_and_fails_loudly_without_gpus():
    """`python bench.py --gpus 2` from a bare shell (no launcher environment) must start two rank processes itself -
    before anything touches a GPU - and report their failure: here there is no GPU, so each rank says so and exits 2
    (the product has no CPU path), and so does the parent. A --gpus that disagrees with WORLD_SIZE is refused."""
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 2, p.stderr
    assert 1 <= p.stderr.count("no GPU visible") <= 2, p.stderr     # (the first rank to fail ends the other one)
    env["WORLD_SIZE"] = "4"
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=120)
    assert p.returncode == 2 and "disagrees with WORLD_SIZE=4" in p.stderr


def test_alignment_pass_changes_encodings_only(amd):
    """tools/align_vop3.py (the pass between `hipcc -S` and the assembler for the two front-end files): its output is its
    input line for line, except that some instructions carry `_e64` where they carried `_e32` or no suffix - nothing
    added, nothing moved, no operand touched - and the shipped library was built from that output."""
    import subprocess
    pkg = ROOT / "opv-cxx-demod_amd"
    for stem in ("k_frontend", "k_frontend_x4"):
        subprocess.run(["make", "-s", "-C", str(pkg), f"build/{stem}.al.s"], check=True, capture_output=True)
        a = (pkg / "build" / f"{stem}.dev.s").read_text().split("\n")
        b = (pkg / "build" / f"{stem}.al.s").read_text().split("\n")
        assert len(a) == len(b)
        widened = 0
        for x, y in zip(a, b):
            if x == y:
                continue
            mx, my = x.split(), y.split()
            assert mx[1:] == my[1:], (x, y)                       # operands untouched
            assert my[0].endswith("_e64") and my[0][:-4] == re.sub(r"_e32$", "", mx[0]), (x, y)
            assert mx[0].startswith("v_")
            widened += 1
        assert widened > 100, stem
        assert (pkg / "build" / f"{stem}.hipfb").stat().st_mtime <= (pkg / "libopv_demod_hip.so").stat().st_mtime


def test_alignment_pass_verifies_its_own_output():
    """the pass compares the re-assembled object with the input instruction by instruction (operation + operands,
    modulo _e32 / _e64 and branch targets) and refuses to write an output that differs"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("align_vop3", ROOT / "opv-cxx-demod_amd" / "tools" / "align_vop3.py")
    al = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(al)
    before = {"k": [[0, 4, "v_add_f32_e32", "v0, v1, v2"], [4, 8, "v_fma_f64", "v[0:1], v[2:3], v[4:5], v[6:7]"],
                    [12, 4, "s_cbranch_scc1", "65531"], [16, 4, "s_endpgm", ""], [20, 4, "s_nop", "0"]]}
    same = {"k": [[0, 8, "v_add_f32_e64", "v0, v1, v2"], [8, 8, "v_fma_f64", "v[0:1], v[2:3], v[4:5], v[6:7]"],
                  [16, 4, "s_cbranch_scc1", "65530"], [20, 4, "s_endpgm", ""]]}
    al.verify(before, same, {"k": [0]})
    for bad in ([[0, 8, "v_add_f32_e64", "v0, v1, v3"]] + same["k"][1:], [[0, 8, "v_sub_f32_e64", "v0, v1, v2"]] + same["k"][1:], same["k"][:-1]):
        with pytest.raises(SystemExit):
            al.verify(before, {"k": bad}, {"k": [0]})


def _replay_interval(ph1, ph2, n_samples):
    """the reference's NCO recurrence (src/opv-mod.cpp:274-279) in numpy scalars: rounded adds, while-loop wraps"""
    pi = np.float64(3.14159265358979323846)
    two_pi = np.float64(2.0) * pi
    inc1 = two_pi * np.float64(-13550.0) / np.float64(2168000.0)
    inc2 = two_pi * np.float64(13550.0) / np.float64(2168000.0)
    a, b = np.float64(ph1), np.float64(ph2)
    for _ in range(n_samples):
        a = a + inc1
        while a > pi: a = a - two_pi
        while a < -pi: a = a + two_pi
        b = b + inc2
        while b > pi: b = b - two_pi
        while b < -pi: b = b + two_pi
    return float(a), float(b)


def test_tx_checkpoint_table_is_the_reference_recurrence(amd):
    """The build-time table the device transmit chain starts from (tools/gen_tx_checkpoints.cpp, embedded by
    csrc/opv_tx_ckpt.cpp): entry 0 is the reset state, every entry is its predecessor advanced by 128 x 40 samples of the
    reference's NCO recurrence - checked at the start, in the middle, across the END of the table (entries 69376 -> 69377
    -> 69378: the last tabulated one and the first two of the host continuation) - bit for bit."""
    ck = amd.tx_checkpoints(0, 6)
    assert ck[0, 0] == 0.0 and ck[0, 1] == 0.0
    for j in range(5):
        assert _replay_interval(ck[j, 0], ck[j, 1], 128 * 40) == (ck[j + 1, 0], ck[j + 1, 1]), j
    n_tab = 4096 * 2168 // 128                       # entries 0 .. n_tab are tabulated (the last one is where longer runs continue)
    for first in (31337, n_tab - 1, n_tab, n_tab + 1):
        a = amd.tx_checkpoints(first, 2)
        assert _replay_interval(a[0, 0], a[0, 1], 128 * 40) == (a[1, 0], a[1, 1]), first
    # and the table agrees with the host modulator's own phase walk (what opv-mod's sha256 pins cover): symbol 128 * 7
    assert tuple(amd.tx_checkpoints(7, 1)[0]) == _replay_interval(0.0, 0.0, 7 * 128 * 40)
