#!/usr/bin/env python3
"""Evidence for the bucketed gradient exchange overlapping the backward pass (data parallel, 2 ranks).

    worker mode (under torchrun, 2 ranks sharing GPU 0, gloo process group -- the code path RCCL takes on a multi-GPU node:
    bucket hook on the plan's comm stream -> async all_reduce -> wait before the optimiser):
        python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/dp_overlap.py worker B steps
    analysis mode (rocprofv3 --kernel-trace --memory-copy-trace CSVs of ONE rank):
        tools/dp_overlap.py analyse <kernel_trace.csv> <memory_copy_trace.csv>

The analysis prints, per step, when each bucket's device->host staging copy (gloo moves CUDA tensors through pinned host memory;
RCCL would launch its ring kernel at the same point of the comm stream) started and ended relative to the backward's GEMM chain.
"""
import csv
import os
import re
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(B, steps):
    import torch
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    import awr_amd
    import awr_oracle as O
    from awr_amd.trainer import TrainEngine
    net = awr_amd.get_deconv_net(18, 14, 2).cuda()
    eng = TrainEngine(net, B, 128, 1.0, coord_weight=0.0, lr=1e-3, process_group=torch.distributed.group.WORLD, autotune=False)
    img, jt = O.synth_batch(B, 128, 14, seed=5 + rank)
    img, jt = img.cuda(), jt.cuda()
    for _ in range(steps):
        eng.step(img, jt)
    torch.cuda.synchronize()
    if rank == 0:
        print("buckets (arena floats lo, hi, ready after backward op):", eng.plan.buckets, flush=True)
    torch.distributed.destroy_process_group()


def short(name):
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", name)).replace("awr::", "")[:48]


def analyse(kfile, mfile):
    ks = []
    for r in csv.DictReader(open(kfile)):
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    ks.sort()
    ms = []
    for r in csv.DictReader(open(mfile)):
        d = r.get("Direction", r.get("Kind", ""))
        ms.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), d))
    ms.sort()
    adam = [i for i, k in enumerate(ks) if k[2].startswith("adam_kernel")]
    if len(adam) < 3:
        print("not enough steps")
        return
    for s in (len(adam) - 3, len(adam) - 2):      # (the last step of a run ends with the process-group teardown)
        seg = ks[adam[s - 1] + 1: adam[s] + 1]
        t0 = seg[0][0]
        loss = [k for k in seg if k[2].startswith("dense_loss_kernel")]
        bwd0 = loss[-1][1] if loss else t0
        chain = [k for k in seg if k[0] >= bwd0 and (k[2].startswith("conv_gemm_kernel") or k[2].startswith("conv_wgrad") or k[2].startswith("stem_bwd"))]
        bwd1 = max(k[1] for k in chain)
        unp = [k for k in seg if k[2].startswith("unpack_batched_kernel")]
        print("step %d: backward GEMM chain %.3f .. %.3f ms after the step's first kernel; optimiser at %.3f ms" %
              (s, (bwd0 - t0) / 1e6, (bwd1 - t0) / 1e6, (seg[-1][0] - t0) / 1e6))
        for i, u in enumerate(unp):
            print("   bucket %d scatter (unpack_batched) %.3f .. %.3f ms" % (i, (u[0] - t0) / 1e6, (u[1] - t0) / 1e6))
        # gloo stages CUDA tensors through pinned host memory: blit kernels (__amd_rocclr_copyBuffer) or SDMA copies
        big = [(k[0], k[1], "blit kernel") for k in seg if k[0] >= bwd0 and k[2].startswith("__amd_rocclr_copyBuffer") and k[1] - k[0] > 100000]
        big += [(m[0], m[1], m[2].replace("MEMORY_COPY_", "")) for m in ms if bwd0 <= m[0] <= seg[-1][1] and m[1] - m[0] > 100000]
        for m in sorted(big):
            inside = "INSIDE the backward" if m[0] < bwd1 else "after the backward"
            print("   staging copy (%s) %.3f .. %.3f ms  (%s: %.3f ms of GEMM chain still to run)" %
                  (m[2], (m[0] - t0) / 1e6, (m[1] - t0) / 1e6, inside, max(0.0, (bwd1 - m[0]) / 1e6)))


if __name__ == "__main__":
    if sys.argv[1] == "worker":
        worker(int(sys.argv[2]), int(sys.argv[3]))
    else:
        analyse(sys.argv[2], sys.argv[3])
