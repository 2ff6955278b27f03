"""Development tool (GPU): the speculative schedule of the border chain against the plain chain schedule, iteration by
iteration (labels, energy, bound bit for bit), also with the runner's rows / labels corrupted on purpose
(STEREO_HIP_TRWS_DEBUG 16384 / 32768: every such segment must be walked a second time and nothing may change).
usage: spec_check.py [iters=6]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from bench import synthetic_volume
from helpers import grid_conn
from stereo_amd.trws import TrwsPlan
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)

def volume(kind, H, W, K):
    if kind == "teddy":
        from stereo_amd import terms as T
        g = np.load(os.path.join(ROOT, "tests", "golden", "teddy_pair.npz"))
        im0, im1 = g["im0"].astype(np.float64), g["im1"].astype(np.float64)
        ncc = T.ncc_volume(im0, im1, np.arange(K, dtype=np.float64), 2, layout=1)
        return np.ascontiguousarray(40.0 * (1.0 - ncc.T))
    return synthetic_volume(H, W, K, seed=3)

def run(env, unary, H, W, K, tol, iters, step=1.0, weights=None):
    for k in ("STEREO_HIP_TRWS_SPEC", "STEREO_HIP_TRWS_DEBUG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    conn = grid_conn(H, W); E = conn.shape[0]; N = H * W
    plan = TrwsPlan(1, K, N, conn.T)
    d_unary = torch.from_numpy(unary).to(dev)
    d_alpha = torch.from_numpy(weights if weights is not None else np.ones(E)).to(dev)
    d_pos = (torch.arange(K, dtype=torch.float64, device=dev) * step)
    plan.bind_device(d_unary.data_ptr(), d_alpha.data_ptr(), tol, d_positions=d_pos.data_ptr(), keepalive=(d_unary, d_alpha, d_pos))
    out = []
    t = time.perf_counter()
    for it in range(iters):
        plan.iterate(1, max_relgap=-1e300)
        lab, en, lb, _ = plan.result(want_labels=True)
        out.append((lab.copy(), en, lb))
    dt = time.perf_counter() - t
    st = plan.spec_stats()
    ser = plan.serial_messages()
    plan.close()
    return out, st, ser, dt

bad = 0
cases = [("noise", 40, 50, 16, 4.0, 1.0), ("noise", 33, 47, 60, 8.0, 0.5), ("noise", 64, 64, 7, 2.0, 1.0), ("teddy", 375, 450, 60, 8.0, 1.0)]
for kind, H, W, K, tol, step in cases:
    unary = volume(kind, H, W, K)
    rng = np.random.default_rng(5)
    E = grid_conn(H, W).shape[0]
    for wname, weights in (("unit weights", None), ("random weights", None if kind == "teddy" else rng.uniform(0.5, 2.0, E))):
        if wname == "random weights" and weights is None: continue
        ref, st0, ser0, dt0 = run({"STEREO_HIP_TRWS_SPEC": "0"}, unary, H, W, K, tol, iters, step, weights)
        for name, env in (("speculative", {}), ("corrupted rows", {"STEREO_HIP_TRWS_DEBUG": "16384"}), ("corrupted rows + labels", {"STEREO_HIP_TRWS_DEBUG": "49152"})):
            got, st, ser, dt = run(env, unary, H, W, K, tol, iters, step, weights)
            same = all(np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2] for a, b in zip(ref, got))
            print("%s %dx%dx%d tol %g step %g %s | %s: %s  stats %s  serial %d (plain %d)  %.1f ms/iter (plain %.1f)" % (
                kind, W, H, K, tol, step, wname, name, "EQUAL" if same else "DIFFERENT", st, ser, ser0, dt / iters * 1e3, dt0 / iters * 1e3), flush=True)
            if not same:
                bad += 1
                for i, (a, b) in enumerate(zip(ref, got)):
                    print("   iter %d: labels differ at %d nodes, energy %r vs %r, bound %r vs %r" % (i + 1, int((a[0] != b[0]).sum()), a[1], b[1], a[2], b[2]))
            if name != "speculative" and st["active"] and st["second_walks"] == 0:
                print("   (no second walk although the runner's output was corrupted)"); bad += 1
print("FAILED %d" % bad if bad else "ALL EQUAL")
sys.exit(1 if bad else 0)
