"""Development tool: time TRW-S iterations of a synthetic volume for a given kernel / size.
usage: time_trws.py [kernel=1] [H=375] [W=450] [K=60] [tol=8] [iters=10] [general=0] [volume=noise|ncc|teddy] [index_order=0] [minplus=0]
general=1: per-edge positions q != qprim (label k + jitter), as a fusion of K plane proposals has them."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from bench import synthetic_volume
from helpers import grid_conn
from stereo_amd.trws import TrwsPlan
a = sys.argv[1:]
kernel = int(a[0]) if len(a) > 0 else 1
H = int(a[1]) if len(a) > 1 else 375
W = int(a[2]) if len(a) > 2 else 450
K = int(a[3]) if len(a) > 3 else 60
tol = float(a[4]) if len(a) > 4 else 8.0
iters = int(a[5]) if len(a) > 5 else 10
general = int(a[6]) if len(a) > 6 else 0
volume = a[7] if len(a) > 7 else "noise"
index_order = int(a[8]) if len(a) > 8 else 0   # 1: STEREO_TRWS_ORDER_INDEX (not the gateway's node order)   # "ncc": NCC cost volume of a synthetic pair (bench.py's workload)
dev = torch.device("cuda", 0)
conn = grid_conn(H, W); E = conn.shape[0]; N = H * W
if volume in ("ncc", "teddy"):
    from bench import synthetic_pair
    from stereo_amd import terms as T
    if volume == "teddy":   # the reference's Teddy pair (375 x 450)
        g = np.load(os.path.join(ROOT, "tests", "golden", "teddy_pair.npz"))
        im0, im1 = g["im0"].astype(np.float64), g["im1"].astype(np.float64)
        H, W = im0.shape[:2]; conn = grid_conn(H, W); E = conn.shape[0]; N = H * W
    else:
        im0, im1 = synthetic_pair(H, W, K, seed=0)
    ncc = T.ncc_volume(im0, im1, np.arange(K, dtype=np.float64), 2, layout=1)
    d_unary = torch.from_numpy(np.ascontiguousarray(40.0 * (1.0 - ncc.T))).to(dev)
else:
    d_unary = torch.from_numpy(synthetic_volume(H, W, K, seed=1)).to(dev)
minplus = int(a[9]) if len(a) > 9 else 0       # 1: STEREO_TRWS_MESSAGES_MINPLUS (plain min-plus messages)
plan = TrwsPlan(kernel, K, N, conn.T, message_mode=(0x100 if index_order else 0) | (1 if minplus else 0))
d_alpha = torch.ones(E, dtype=torch.float64, device=dev)
d_pos = torch.arange(K, dtype=torch.float64, device=dev)
if general:
    g = torch.Generator(device=dev); g.manual_seed(5)
    d_q = torch.arange(K, dtype=torch.float64, device=dev)[None, :] + 0.25 * torch.rand(E, K, dtype=torch.float64, device=dev, generator=g)
    d_qp = torch.arange(K, dtype=torch.float64, device=dev)[None, :] + 0.25 * torch.rand(E, K, dtype=torch.float64, device=dev, generator=g)
    plan.bind_device(d_unary.data_ptr(), d_alpha.data_ptr(), tol, d_q=d_q.data_ptr(), d_qprim=d_qp.data_ptr(), keepalive=(d_unary, d_alpha, d_q, d_qp))
else:
    plan.bind_device(d_unary.data_ptr(), d_alpha.data_ptr(), tol, d_positions=d_pos.data_ptr(), keepalive=(d_unary, d_alpha, d_pos))
plan.iterate(2, max_relgap=-1e300)
plan.serial_messages(reset=True)
torch.cuda.synchronize(); t = time.perf_counter()
plan.iterate(iters, max_relgap=-1e300)
dt = (time.perf_counter() - t) / iters
_, en, lb, it = plan.result(want_labels=False)
print("kernel %d %dx%dx%d tol %g: path %d, %.2f ms/iter (%.1f it/s), serial messages %d, energy %.6f lb %.6f" % (
    kernel, W, H, K, tol, plan.path(), dt * 1e3, 1 / dt, plan.serial_messages(), en, lb))
if hasattr(plan, "spec_stats"): print("spec", plan.spec_stats())
plan.close()
