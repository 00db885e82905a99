"""dev: wall time of the drop-in CLI on a 1000-frame capture, from a file and through a pipe, for one or more binaries
(names under opv-cxx-demod_amd/bin), plus the fixed cost of a run on empty input."""
import hashlib, os, subprocess, sys, time
from pathlib import Path
B = Path(__file__).resolve().parents[2] / "opv-cxx-demod_amd" / "bin"
cap = "/tmp/c1000.iq"
with open(cap, "wb") as f:
    subprocess.run([str(B / "opv-mod"), "-S", "W5NYV", "-B", "1000"], stdout=f, check=True)
n = os.path.getsize(cap) // 4
for name in sys.argv[1:] or ["opv-demod"]:
    for rep in range(3):
        t0 = time.time()
        with open(cap, "rb") as f:
            a = subprocess.run([str(B / name), "-s", "-r", "-q"], stdin=f, capture_output=True).stdout
        t1 = time.time()
        p = subprocess.Popen(["cat", cap], stdout=subprocess.PIPE)
        b = subprocess.run([str(B / name), "-s", "-r", "-q"], stdin=p.stdout, capture_output=True).stdout
        p.wait()
        t2 = time.time()
        print(f"{name} rep {rep}: file {t1 - t0:.3f} s = {n / (t1 - t0) / 1e6:.1f} MS/s, pipe {t2 - t1:.3f} s = {n / (t2 - t1) / 1e6:.1f} MS/s, "
              f"{len(a)} / {len(b)} bytes, sha {hashlib.sha256(b).hexdigest()[:16]}", flush=True)
t0 = time.time()
subprocess.run([str(B / "opv-demod"), "-s", "-r", "-q"], stdin=subprocess.DEVNULL, capture_output=True)
print(f"empty input (process start + HIP init + create / destroy): {time.time() - t0:.3f} s")
