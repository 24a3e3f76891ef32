import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import soundml_amd as S
from soundml_amd import Stft
from oracle import soundml_oracle as O
# find the failing draw by replaying the generator of tools/fuzz_parity.py up to it
import importlib.util, types
src = open("tools/fuzz_parity.py").read()
# minimal replay: run feature_case until the failure prints; capture inputs by monkeypatching griffin_lim
calls = {}
orig = Stft.griffin_lim
def spy(c, mag, **kw):
    calls["last"] = (c, mag, kw)
    return orig(c, mag, **kw)
Stft.griffin_lim = spy
sys.argv = ["fuzz", "1500", "95", "features"]
ns = {"__name__": "fuzzmod", "__file__": os.path.join(os.getcwd(), "tools", "fuzz_parity.py"), "calls": calls}
code = src.replace("sys.exit(min(fails, 100))", "raise SystemExit")
try:
    exec(compile(code.replace("print(\"FAIL\", params", "globals()['failed'] = dict(params); calls['fail'] = calls.get('last'); print(\"FAIL\", params"), "fuzz", "exec"), ns)
except SystemExit:
    pass
c, mag, kw = calls["fail"]
o = O.stft_config(512, hop=c.hop)
want = O.griffin_lim(o, mag, n_iter=kw["n_iter"], momentum=kw["momentum"], init=kw["init"])
for interior in ("float32", "float64"):
    S.set_interior(interior)
    got = orig(c, mag, **kw)
    e = np.abs(got - want)
    print(interior, "rel l2", np.linalg.norm(got - want) / np.linalg.norm(want), "max abs", e.max(), "peak", np.abs(want).max(),
          "samples above 1e-3 of peak:", int((e > 1e-3 * np.abs(want).max()).sum()), "of", e.size)
S.set_interior("float32")
# sensitivity: perturb the magnitudes by one float32 ulp and rerun the ORACLE
mag2 = np.nextafter(mag, np.float32(np.inf))
w2 = O.griffin_lim(o, mag2, n_iter=kw["n_iter"], momentum=kw["momentum"], init=kw["init"])
print("oracle vs oracle with magnitudes moved by one ulp: rel l2", np.linalg.norm(w2 - want) / np.linalg.norm(want))
