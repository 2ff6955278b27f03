"""Development tool (CPU): how often does a border-chain message of the Teddy volume differ from plain min-plus?
The speculative chain runner hands min-plus rows ahead of the certified visits; a row that differs
from the reference's envelope result costs a squash.  Uses the oracle's diagnostic mode 2 + its per-message record.
usage: census_chain.py [iters=6] [volume=teddy|noise]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from helpers import grid_conn
from oracle import pyoracle as po, terms as ot
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
volume = sys.argv[2] if len(sys.argv) > 2 else "teddy"
K = 60
if volume == "teddy":
    g = np.load(os.path.join(ROOT, "tests", "golden", "teddy_pair.npz"))
    im0, im1 = g["im0"].astype(np.float64), g["im1"].astype(np.float64)
    H, W = im0.shape[:2]
    ncc = ot.compute_ncc(im0, im1, np.arange(K, dtype=np.float64), 2)
    unary = np.ascontiguousarray(40.0 * (1.0 - ncc.reshape(H * W, K, order="F")))
else:
    from bench import synthetic_volume
    H, W = 375, 450
    unary = synthetic_volume(H, W, K, seed=1)
N = H * W
conn = grid_conn(H, W); E = conn.shape[0]
S = po.trws_structure(N, conn)
rank = S["rank"]; order = np.argsort(rank)
per_iter = 2 * E
buf = np.zeros(iters * per_iter, np.uint8)
po.lib().oracle_trws_message_trace(buf.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int64(buf.size))
pos = np.arange(K, dtype=np.float64)
q = np.broadcast_to(pos, (E, K)).copy()
lab, en, lb, it = po.trws(1, unary, conn, q, q, np.ones(E), 8.0, maxiter=iters, max_relgap=-1e300, mode=2)
po.lib().oracle_trws_message_trace(None, C.c_int64(0))
# sequence -> (sweep, source rank, destination rank)
seq_src = np.zeros(per_iter, np.int64); seq_dst = np.zeros(per_iter, np.int64)
n = 0
for r in range(N):
    i = order[r]
    for k in range(S["fwd_ptr"][i], S["fwd_ptr"][i + 1]):
        e = S["fwd_idx"][k]; seq_src[n] = r; seq_dst[n] = rank[S["head"][e]]; n += 1
for r in range(N - 1, -1, -1):
    i = order[r]
    for k in range(S["bwd_ptr"][i], S["bwd_ptr"][i + 1]):
        e = S["bwd_idx"][k]; seq_src[n] = r; seq_dst[n] = rank[S["tail"][e]]; n += 1
assert n == per_iter
C_ = 2 * (H + W) - 4
chain = (seq_src < C_) & (seq_dst < C_) & (np.abs(seq_src - seq_dst) == 1)
serial_part = chain & (np.maximum(seq_src, seq_dst) < C_ - (W - 1))
print("chain nodes %d, in-run chain messages per iteration %d (twins counted twice)" % (C_, chain.sum()))
for t in range(iters):
    b = buf[t * per_iter:(t + 1) * per_iter]
    for name, m in (("fwd", np.arange(per_iter) < E), ("bwd", np.arange(per_iter) >= E)):
        c = chain & m
        neq = (b[c] & 1).sum(); cert = ((b[c] >> 1) & 1).sum()
        where = np.unique(np.minimum(seq_src, seq_dst)[c][(b[c] & 1) == 1])
        alln = (b[m] & 1).sum(); allc = ((b[m] >> 1) & 1).sum()
        print("iter %d %s: chain messages %d, certificate fails %d, differ from min-plus %d (at %d chain nodes: %s) | whole sweep: cert fails %d, differ %d"
              % (t + 1, name, c.sum(), cert, neq, len(where), where[:12], allc, alln))
