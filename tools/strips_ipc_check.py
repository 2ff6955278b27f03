#!/usr/bin/env python
"""Two (or more) PROCESSES, one strip each, hand-over through HIP IPC mappings of each other's
arrays -- the multi-GPU code path of stereo_amd.strips.TrwsStripRank -- on whatever devices are
there (all ranks share device 0 on a 1-GPU box: the peer stores then stay inside one HBM, but the
IPC export / open, the flag protocol across processes and the reduction are the real ones).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
      --master-port 29533 tools/strips_ipc_check.py [H W K]
Rank 0 compares with the single-plan result and prints IPC_STRIPS_OK."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    H, W, K = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (40, 46, 16)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(ndev, 1)
    backend = "nccl" if ndev >= world else "gloo"   # RCCL refuses two ranks on one GPU
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo")
    from stereo_amd import _lib
    from stereo_amd.strips import TrwsStripRank
    from stereo_amd.trws import TrwsPlan
    from helpers import trws_problem
    _lib.lib().stereo_hip_set_device(local)
    kind = "fronto" if K > 64 else "general"
    dev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")
    conn = trws_problem(7, H, W, K, kind=kind)["conn"]
    s = TrwsStripRank(1, K, H, W, conn.T, rank, world, dist, dev,
                      max_workgroups=max(2, (int(_lib.lib().stereo_hip_device_cus()) or 256) // world) if ndev < world else 0)
    ok = True
    iters = 4
    # two problems one after the other on the same strips: the second upload resets arrays the
    # neighbouring rank writes into (TrwsStripRank brackets it with barriers)
    for seed in (7, 8):
        p = trws_problem(seed, H, W, K, kind=kind)
        if kind == "fronto":
            s.upload(p["unary"].T, p["alphas"], 8.0, positions=np.arange(K, dtype=np.float64))
        else:
            s.upload(p["unary"].T, p["alphas"], 3.0, q=p["q"].T, qprim=p["qprim"].T)
        done, _ = s.iterate(iters, max_relgap=-1e300)
        assert done == iters
        idx, lab = s.own_labels()
        parts = [None] * world
        dist.all_gather_object(parts, (idx, lab, local))
        if rank == 0:
            full = np.zeros(H * W)
            for i, l, _ in parts:
                full[i] = l
            one = TrwsPlan(1, K, H * W, p["conn"].T)
            if kind == "fronto":
                one.upload(p["unary"].T, p["alphas"], 8.0, positions=np.arange(K, dtype=np.float64))
            else:
                one.upload(p["unary"].T, p["alphas"], 3.0, q=p["q"].T, qprim=p["qprim"].T)
            one.iterate(iters, max_relgap=-1e300)
            lab1, en1, lb1, _ = one.result()
            one.close()
            good = bool(np.array_equal(full, lab1)) and abs(s.energy - en1) <= 1e-12 * abs(en1) and abs(s.lb - lb1) <= 1e-12 * abs(lb1)
            ok = ok and good
            print("strips %d backend %s devices %s problem %d labels_equal %s energy %.10f vs %.10f lb %.10f vs %.10f" %
                  (world, backend, [d for _, _, d in parts], seed, np.array_equal(full, lab1), s.energy, en1, s.lb, lb1))
    if rank == 0:
        print("IPC_STRIPS_OK" if ok else "IPC_STRIPS_MISMATCH")
    dist.barrier()
    s.close()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
