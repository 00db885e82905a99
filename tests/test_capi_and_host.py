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
    assert L.opv_abi_version() == 7


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
    # the reference's usage text (tests/golden/usage_stderr.txt: made by its binary, ref src/opv-demod.cpp:962-971), line for
    # line, THEN the two flags the reference does not have (INTEGRATION.md lists them)
    ref = (ROOT / "tests" / "golden" / "usage_stderr.txt").read_text().rstrip("\n").split("\n")
    got = r.stderr.decode().rstrip("\n").split("\n")
    assert got[0].startswith("Usage: ") and got[0].endswith(" [options] < input.iq")
    assert got[1:len(ref)] == ref[1:]
    assert [ln.split()[0] for ln in got[len(ref):]] == ["--device", "--capacity-sec"]
    live = ROOT / "oracle" / "_ref" / "opv-demod"
    if live.exists():
        lr = subprocess.run([str(live), "-h"], capture_output=True).stderr.decode().rstrip("\n").split("\n")
        assert lr[1:] == ref[1:]
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


def test_tx_stream_any_split_of_a_run_equals_one_call(amd, oracle):
    """opv_tx_stream_*: the modulator's state (two NCO phases, differential sign, symbol parity: HDLModulator,
    src/opv-mod.cpp:219-291) carried from call to call - any split of a run into calls, empty calls included, gives the
    bytes of one opv_tx_modulate call (= the oracle's, = opv-mod's by the sha256 pins above); reset starts a new run."""
    rng = np.random.default_rng(77)
    fr = rng.integers(0, 256, (9, 134), dtype=np.uint8)
    whole = amd.modulate(fr)
    assert np.array_equal(whole, oracle.modulate(fr))
    for split in ([9], [1] * 9, [2, 0, 3, 4], [8, 1]):
        st = amd.TxStream()
        parts, at = [], 0
        for n in split:
            parts.append(st.frames(fr[at:at + n]))
            at += n
        parts.append(st.tail())
        assert np.array_equal(np.concatenate(parts), whole), split
        st.reset()                                                   # a second run on the same object (opv-mod -c, :506)
        assert np.array_equal(np.concatenate([st.frames(fr[:2]), st.tail()]), amd.modulate(fr[:2]))
        st.close()


def test_tx_frame_taps_against_the_oracle(amd, oracle):
    """opv_tap_tx_frame: randomiser, encoder and interleaver outputs of a frame (src/opv-mod.cpp:158-213)"""
    rng = np.random.default_rng(78)
    perm = oracle.deinterleave_perm()                                # decoder side: deint[i] = received[perm[i]]
    for k in range(5):
        f = rng.integers(0, 256, 134, dtype=np.uint8)
        r, coded, inter = amd.tx_frame_taps(f)
        assert np.array_equal(r, f ^ oracle.lfsr_table())
        assert np.array_equal(inter, oracle.encode_frame(f))
        assert np.array_equal(coded, inter[perm])


def _run_opv_mod(binary, argv, data, take):
    import subprocess
    if take is None:
        p = subprocess.run([str(binary)] + argv, input=data, capture_output=True, timeout=300)
        return p.returncode, p.stderr.decode().replace(str(binary), "PROG"), p.stdout
    p = subprocess.Popen([str(binary)] + argv, stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out = b""
    while len(out) < take:
        b = p.stdout.read(take - len(out))
        if not b:
            break
        out += b
    p.stdout.close()
    p.kill()
    p.wait()
    return None, None, out


def test_opv_mod_process_contract_vs_reference_fixture(amd):
    """bin/opv-mod against what the reference's opv-mod printed and wrote for the same command lines
    (tests/golden/opv_mod_cli.json, made by tests/golden/make_golden_opv_mod.py from oracle/_ref/opv-mod): exit status,
    stdout bytes, and ALL of stderr - the -v banner and per-frame debug lines (src/opv-mod.cpp:171-183,198-209,326-329,
    456-469), progress lines, warnings (partial frame :377, long callsign :452), the three mode errors (:432-447) and the
    usage text, which may differ in the one line of the flag that is ours (-G). -c (continuous, :503-524): the first
    1.7 MB, i.e. two and a half passes over the two BERT frames with the modulator reset between passes."""
    import json
    import sys
    sys.path.insert(0, str(ROOT / "tests" / "golden"))
    from make_golden_opv_mod import stdin_bytes
    fix = json.loads((ROOT / "tests" / "golden" / "opv_mod_cli.json").read_text())
    ours = ROOT / "opv-cxx-demod_amd" / "bin" / "opv-mod"
    assert len(fix) >= 11
    for name, c in fix.items():
        rc, err, out = _run_opv_mod(ours, c["argv"], stdin_bytes(c["stdin"]), c["take"])
        assert len(out) == c["stdout_len"] and hashlib.sha256(out).hexdigest() == c["stdout_sha256"], name
        if c["take"] is not None:
            continue
        assert rc == c["rc"], name
        mine = [ln for ln in err.split("\n") if not ln.startswith("  -G DEVICE")]
        assert mine == c["stderr"].split("\n"), name


def test_opv_mod_raw_mode_writes_as_frames_arrive(amd):
    """`opv-mod -R` on a live source (the reference modulates and writes frame by frame, src/opv-mod.cpp:479-493): the samples
    of frame k can be read before frame k + 1 is written, and the whole output equals the one-shot run's."""
    import os
    import subprocess
    ours = str(ROOT / "opv-cxx-demod_amd" / "bin" / "opv-mod")
    fr = [bytes([17 * k + 3] * 134) for k in range(3)]
    p = subprocess.Popen([ours, "-R"], stdin=subprocess.PIPE, stdout=subprocess.PIPE)
    got = b""
    need = 2168 * 40 * 4
    for k in range(3):
        p.stdin.write(fr[k])
        p.stdin.flush()
        chunk = b""
        while len(chunk) < need:                                     # blocks forever (-> test timeout) if the CLI waited for EOF
            b = os.read(p.stdout.fileno(), need - len(chunk))
            assert b, "opv-mod closed its output early"
            chunk += b
        got += chunk
    p.stdin.close()
    got += p.stdout.read()
    assert p.wait() == 0
    assert got == subprocess.run([ours, "-R"], input=b"".join(fr), capture_output=True, timeout=120).stdout
    assert np.array_equal(np.frombuffer(got, np.int16), amd.modulate(np.frombuffer(b"".join(fr), np.uint8).reshape(3, 134)))


def test_flat_top_zones_hold_for_this_libm():
    """k_tx_modulate.hip decides the flat tops of the NCOs (sin or cos = 1 - eps^2/2 at every symbol start) by the symbol
    index: exactly +/-1.0 while the phase drift eps < 0.90e-8, below 1.0 from 1.25e-8 on, libm's own answer in between.
    That is a statement about THIS machine's libm (the one the host modulator and the reference use): scanned here, both
    zones, both functions, all four quadrant points."""
    libm = C.CDLL("libm.so.6")
    for f in (libm.sin, libm.cos):
        f.restype = C.c_double
        f.argtypes = [C.c_double]
    pi = 3.14159265358979323846
    tops = [(libm.sin, pi / 2), (libm.sin, -pi / 2), (libm.cos, 0.0), (libm.cos, pi), (libm.cos, -pi)]
    rng = np.random.default_rng(5)
    exact = np.concatenate([np.linspace(0.0, 0.90e-8, 4001), rng.uniform(0.0, 0.90e-8, 4000)])
    below = np.concatenate([np.linspace(1.25e-8, 4e-8, 4001), 10.0 ** rng.uniform(np.log10(1.25e-8), -5.0, 4000)])
    for f, x0 in tops:
        for e in exact:
            assert abs(f(x0 + e)) == 1.0 and abs(f(x0 - e)) == 1.0, (x0, e)
        for e in below:
            assert abs(f(x0 + e)) < 1.0 and abs(f(x0 - e)) < 1.0, (x0, e)


def test_offset_tie_host_evaluation_is_the_reference_evaluation(oracle, tmp_path):
    """csrc/opv_offset_host.cpp (the product's host-side decision of offset-search near-ties) against the oracle's
    estimate_offset (ref src/opv-demod.cpp:131-202), on this machine's libm: the energy of every coarse candidate EQUAL bit for bit
    on a noisy MSK opening and on a real-valued (exactly tying) one; the pinned probe energy is the oracle's table entry for
    the probe sequence; and the whole decision, fed a polynomial whose values tie at the edges, keeps the reference's first
    maximum."""
    import subprocess
    pkg = ROOT / "opv-cxx-demod_amd"
    so = tmp_path / "libtie.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-o", str(so), str(pkg / "csrc" / "opv_offset_host.cpp")], check=True)
    syms = subprocess.run(["nm", "-D", "--defined-only", str(so)], capture_output=True, text=True, check=True).stdout.split()
    name = lambda part: next(x for x in syms if part in x)
    L = C.CDLL(str(so))
    energy = getattr(L, name("opv_offset_candidate_energy"))
    energy.restype = C.c_double
    energy.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_bool]
    probe = getattr(L, name("opv_offset_host_libm_matches_reference"))
    probe.restype = C.c_bool
    decide = getattr(L, name("opv_offset_decide_on_host"))
    decide.restype = C.c_double
    decide.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_bool]
    assert probe()                                               # this image's glibc gives the pinned energy
    # the probe sequence and its pinned energy (csrc/opv_offset_host.h) against the oracle's table
    v, seq = 0x2545F491, np.empty(80000, np.int16)
    for k in range(80000):
        v = (v * 1664525 + 1013904223) & 0xFFFFFFFF
        seq[k] = (v >> 16) % 4001 - 2000
    hdr = (pkg / "csrc" / "opv_offset_host.h").read_text()
    pinned = float.fromhex(re.search(r"OPV_OFFSET_PROBE_ENERGY (0x[0-9a-fp.+]+)", hdr).group(1))
    _, e = oracle.estimate_offset(seq, energies=True)
    assert e[117] == pinned == energy(seq.ctypes.data, 1000, 1425.0, True) == energy(seq.ctypes.data, 1000, 1425.0, False)
    from oracle_lib import impair
    iq = oracle.modulate(oracle.bert_frames(1))
    noisy = np.ascontiguousarray(impair(iq[: 2 * 30000], amp=1500.0, f0_hz=640.0, ebn0_db=9.0, seed=3))
    real = np.zeros(2 * 40000, np.int16)
    real[0::2] = np.rint(9000 * np.cos(2 * np.pi * 36000.0 * np.arange(40000) / 2168000.0 + 0.3))
    for x in (noisy, real):
        off, e = oracle.estimate_offset(x, energies=True)
        nsym = min(x.size // 2, 40000) // 40
        mine = np.array([energy(x.ctypes.data, nsym, -1500.0 + 25.0 * c, bool(c & 1)) for c in range(121)])   # (threaded / not: same numbers)
        assert np.array_equal(mine, e[:121])
    assert e[0] == e[120] and off == -1530.0                     # the real-valued capture: an exact mirror tie, first maximum kept
    # the decision: a polynomial in theta whose values at the edge candidates are the winners and equal (even, convex)
    poly = np.zeros(19)
    poly[0], poly[2] = e[60], (e[0] - e[60]) / (2 * np.pi * 1500.0 / 2168000.0) ** 2
    out, ties = np.zeros(134), C.c_uint32(0)
    power = float(np.sum(real.astype(np.float64) ** 2))        # sum |x|^2 over the samples used (exact in fp64)
    for threads in (True, False):
        est = decide(real.ctypes.data, 1000, poly.ctypes.data, power, out.ctypes.data, C.byref(ties), threads)
        assert est == off and ties.value >= 2 and out[0] == e[0] and out[120] == e[120]
    # the power-scaled band: the same polynomial with its edge values 1e-9 (relative) apart is a clear decision for a signal the
    # tones match (energy ~ 40 x power: nothing is re-evaluated) and a near-tie for one whose correlation is weak against its
    # power (energy = 1e-8 x 40 x power: the band is 1e4 times wider)
    poly2 = poly.copy()
    poly2[1] = 1e-9 * e[0] / (2 * (2 * np.pi * 1500.0 / 2168000.0))   # odd term: E(+1500) - E(-1500) = 1e-9 E
    decide(real.ctypes.data, 1000, poly2.ctypes.data, e[0] / 40.0, out.ctypes.data, C.byref(ties), True)
    assert ties.value == 0
    decide(real.ctypes.data, 1000, poly2.ctypes.data, e[0] * 1e8 / 40.0, out.ctypes.data, C.byref(ties), True)
    assert ties.value >= 2


def test_rx_bridge_rejects_a_malformed_device_list(amd):
    """`opv-rx-bridge --devices x,y` used to spin forever (strtol does not advance on a non-number); every malformed list is
    refused with a message and exit status 2 before anything touches a GPU."""
    import subprocess
    b = str(ROOT / "opv-cxx-demod_amd" / "bin" / "opv-rx-bridge")
    for bad in ("x,y", "0,,1", "-1", ",0", "0;1", ""):
        p = subprocess.run([b, "--devices", bad, "/dev/null"], capture_output=True, stdin=subprocess.DEVNULL, timeout=20)
        assert p.returncode == 2 and b"--devices takes a comma-separated list" in p.stderr, (bad, p.stderr)


def test_product_library_carries_the_product_front_ends_only():
    """The shipped libopv_demod_hip.so exports exactly the front-end kernels opv_process can launch - one wave per stream
    (two launch shapes), four streams per wave (one), sixteen per wave (three); each has an oracle parity test."""
    import subprocess
    so = ROOT / "opv-cxx-demod_amd" / "libopv_demod_hip.so"
    out = subprocess.run(["nm", "-D", "--defined-only", str(so)], capture_output=True, text=True, check=True).stdout
    kernels = sorted(ln.split()[-1] for ln in out.splitlines() if ln.split()[-1].startswith("k_msk_frontend"))
    assert kernels == ["k_msk_frontend_rb", "k_msk_frontend_rb_wg4", "k_msk_frontend_x16", "k_msk_frontend_x16_wg4",
                       "k_msk_frontend_x16_wg8", "k_msk_frontend_x4_wg4"], kernels


def test_cli_refuses_non_finite_flag_values():
    """`opv-demod -s -o inf`: the reference spins forever in its phase-wrap loops (src/opv-demod.cpp:255-262); here the value is
    refused by opv_create before any device is touched - exit status 2 and a message, also on a box without a GPU."""
    import subprocess
    exe = str(ROOT / "opv-cxx-demod_amd" / "bin" / "opv-demod")
    for flags in (["-s", "-o", "inf"], ["-s", "-a", "nan"], ["-c", "-p", "-inf"], ["-s", "-o", "1e999"]):
        p = subprocess.run([exe, "-q"] + flags, input=b"\x00" * 4000, capture_output=True, timeout=30)
        assert p.returncode == 2 and b"must be finite" in p.stderr, (flags, p.returncode, p.stderr)
