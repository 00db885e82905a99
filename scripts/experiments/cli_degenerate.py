"""dev: bin/opv-demod against the reference binary (oracle/_ref/opv-demod) on degenerate inputs - empty, a few bytes, less than a
symbol, less than a chunk, odd byte counts - in -s and batch mode, with and without -q / -r: exit status, stdout and stderr."""
import subprocess, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
ours, ref = str(ROOT / "opv-cxx-demod_amd" / "bin" / "opv-demod"), str(ROOT / "oracle" / "_ref" / "opv-demod")
mod = str(ROOT / "opv-cxx-demod_amd" / "bin" / "opv-mod")
sig = subprocess.run([mod, "-S", "W5NYV", "-B", "2"], capture_output=True).stdout
rng = np.random.default_rng(1)
inputs = {"empty": b"", "1 byte": b"\x01", "3 bytes": b"\x01\x02\x03", "1 sample": b"\x10\x00\x20\x00", "39 samples": sig[:39 * 4], "40 samples": sig[:160],
          "49 samples": sig[:49 * 4], "50 samples": sig[:200], "51 samples + 1 byte": sig[:205], "1000 samples": sig[:4000], "40000 samples": sig[:160000], "40001 samples": sig[:160004],
          "one chunk - 1": sig[: 86719 * 4], "one chunk": sig[: 86720 * 4], "one chunk + 3 bytes": sig[: 86720 * 4 + 3], "noise 5000": rng.integers(-3000, 3000, 10000).astype(np.int16).tobytes(),
          "zeros 100000": bytes(400000)}
bad = 0
for name, data in inputs.items():
    for flags in (["-s"], ["-s", "-r", "-q"], [], ["-r"], ["-q", "-o", "300"], ["-s", "-o", "-250", "-a", "0.002"]):
        a = subprocess.run([ours] + flags, input=data, capture_output=True, timeout=120)
        b = subprocess.run([ref] + flags, input=data, capture_output=True, timeout=120)
        same = a.returncode == b.returncode and a.stdout == b.stdout and a.stderr == b.stderr
        if not same:
            bad += 1
            print("DIFF", name, flags, "rc", a.returncode, b.returncode, "stdout equal", a.stdout == b.stdout, "stderr equal", a.stderr == b.stderr)
            if a.stderr != b.stderr:
                al, bl = a.stderr.decode(errors="replace").split("\n"), b.stderr.decode(errors="replace").split("\n")
                for x, y in zip(al, bl):
                    if x != y:
                        print("   ours:", x[:150]); print("   ref :", y[:150]); break
                if len(al) != len(bl): print("   line counts", len(al), len(bl))
print(len(inputs) * 6, "runs,", bad, "differences")
