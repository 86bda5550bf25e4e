"""Debug driver (not collected by pytest): prints every parity error for the named cases."""
import sys, traceback
import numpy as np
sys.path.insert(0, ".")
from tests import test_gpu_parity as T

names = sys.argv[1:] or list(T.CASES)
for n in names:
    try:
        try:
            errs = T.run_case(n)
            top=sorted(errs.items(), key=lambda kv:-kv[1])[:4]; print(n, "OK max err", top)
        except AssertionError as e:
            print(n, "FAIL", str(e)[:3000])
    except Exception:
        traceback.print_exc()
        break
