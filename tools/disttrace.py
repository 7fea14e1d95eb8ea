import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from parmgmc_amd import GridMCSOR
from parmgmc_amd.dist import IpcSlabDriver, RcclSlabDriver
g = GridMCSOR(512, 512, 64, 10.0)
b = g.to_cvec(torch.ones(g.n, dtype=torch.float64, device="cuda"))
y = g.new_cvec()
drv = IpcSlabDriver(g, 0, 1, loopback=True)
drv.sample_cvec(b, y, 5, True, 1, 1, 0)
torch.cuda.synchronize()
drv.sample_cvec(b, y, 20, True, 1, 1, 5)
torch.cuda.synchronize()
