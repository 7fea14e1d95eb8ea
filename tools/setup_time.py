import time, os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from parmgmc_amd import MGMC
for full in (0, 1):
    if full: os.environ["PMG_MG_FULL_GALERKIN"] = "1"
    t = time.time(); mg = MGMC(257, 257, 257, 10.0, 5); mg.setup(); torch.cuda.synchronize(); print("full" if full else "proxy", "setup 257^3 x5:", round(time.time() - t, 2), "s", flush=True)
    del mg
