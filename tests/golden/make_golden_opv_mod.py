#!/usr/bin/env python3
"""Fixture for the transmit side's process contract: what the REFERENCE's `opv-mod` (oracle/_ref/opv-mod, compiled from
/root/reference by `make -C oracle ref`) prints and writes for a set of command lines - exit status, the whole of stderr
(with the program's path replaced by PROG) and sha256 + length of stdout (for `-c`, which never ends by itself, of the
first `take` bytes). Inputs of the raw-mode cases are made by the seeded recipe in CASES. What is stored is DATA.

  python tests/golden/make_golden_opv_mod.py
"""
import hashlib
import json
import subprocess
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent))
from oracle_lib import ref_binary  # noqa: E402

FRAME_IQ_BYTES = 2168 * 40 * 4


def raw_frames(n, seed):
    return np.random.default_rng(seed).integers(0, 256, (n, 134), dtype=np.uint8).tobytes()


CASES = {   # name: (argv, stdin recipe (n_frames, seed) or None, bytes of stdout to take or None = all)
    "bert12_verbose": (["-S", "W5NYV", "-B", "12", "-v"], None, None),
    "bert3_token": (["-S", "KB5MU-7", "-B", "3", "-t", "0x123456", "-v"], None, None),
    "raw120_verbose": (["-R", "-v"], (120, 11), None),
    "raw_partial_tail": (["-R"], (2, 12, 57), None),                 # two frames + 57 bytes: the partial frame is dropped
    "raw_empty": (["-R", "-v"], (0, 0), None),
    "continuous_2": (["-S", "W5NYV", "-B", "2", "-c"], None, 5 * FRAME_IQ_BYTES + 12344),
    "long_callsign": (["-S", "TOOLONGCALLSIGN", "-B", "1"], None, None),
    "no_mode": ([], None, None),
    "both_modes": (["-R", "-B", "3"], None, None),
    "no_callsign": (["-B", "3"], None, None),
    "unknown_flag": (["-Z"], None, None),
}


def stdin_bytes(recipe):
    if recipe is None:
        return b""
    n, seed, *extra = recipe
    return raw_frames(n, seed) + (bytes(range(extra[0])) if extra else b"")


def run(binary, argv, data, take):
    """(rc, stderr text with the program path as PROG, stdout bytes). take: read that many bytes, then end the child."""
    if take is None:
        p = subprocess.run([str(binary)] + argv, input=data, capture_output=True, timeout=120)
        return p.returncode, p.stderr.decode().replace(str(binary), "PROG"), p.stdout
    p = subprocess.Popen([str(binary)] + argv, stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out = b""
    while len(out) < take:
        b = p.stdout.read(take - len(out))
        if not b:
            break
        out += b
    p.stdout.close()          # the writer ends on SIGPIPE / a failed write
    p.kill()
    p.wait()
    return None, None, out


def main():
    ref = ref_binary("opv-mod")
    assert ref, "run `make -C oracle ref` first"
    meta = {}
    for name, (argv, recipe, take) in CASES.items():
        rc, err, out = run(ref, argv, stdin_bytes(recipe), take)
        meta[name] = {"argv": argv, "stdin": recipe, "take": take, "rc": rc, "stderr": err,
                      "stdout_len": len(out), "stdout_sha256": hashlib.sha256(out).hexdigest()}
        print(name, rc, len(out), (err or "")[:60].replace("\n", " | "))
    (HERE / "opv_mod_cli.json").write_text(json.dumps(meta, indent=1))


if __name__ == "__main__":
    main()
