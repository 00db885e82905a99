"""dev: instruction census of the steady-state symbol loop (the backward branch that spans 120 v_fmac_f64_dpp = four
symbols) of the one-wave front-end kernels, read from the aligned device assembly the library is built from."""
import collections
import re
import sys
from pathlib import Path

asm = Path(__file__).resolve().parents[2] / "opv-cxx-demod_amd" / "build" / "k_frontend.al.s"
L = asm.read_text().split("\n")
for fn in sys.argv[1:] or ("k_msk_frontend_rb", "k_msk_frontend_rd"):
    start = next(i for i, l in enumerate(L) if l.startswith(fn + ":"))
    end = next(i for i in range(start, len(L)) if L[i].startswith(".Lfunc_end"))
    labels = {}
    for i in range(start, end):
        m = re.match(r"^(\.LBB\d+_\d+):", L[i])
        if m:
            labels[m.group(1)] = i
    best = None
    for i in range(start, end):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", L[i])
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a = labels[m.group(1)]
            if sum(1 for k in range(a, i) if "v_fmac_f64_dpp" in L[k]) == 120:
                best = (a, i)
    a, b = best
    body = [l.strip().split(";")[0].strip() for l in L[a:b + 1] if l.strip() and not l.strip().startswith(";") and not re.match(r"^\.L", l.strip())]
    c = collections.Counter(x.split()[0].replace("_e32", "").replace("_e64", "") for x in body)
    print(fn, len(body), "instructions per trip =", len(body) / 4, "per symbol")
    print("  ", sorted(c.items(), key=lambda kv: -kv[1]))
    Path(f"/tmp/loop_{fn}.s").write_text("\n".join(body))
