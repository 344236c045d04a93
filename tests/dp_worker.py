"""Worker of tests/test_dp_gpu.py: one rank of a 2-rank data-parallel TrainEngine run.  Both ranks share GPU 0; the process
group is gloo (it accepts CUDA tensors and stages them through the host), which exercises exactly the code path RCCL takes on a
multi-GPU node: rank-0 broadcast of parameters / BN buffers, bucket hooks inside the backward, async all-reduce + wait, 1/world
in the optimiser."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))


def main():
    rank, world, out = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[1]
    backend = os.environ.get("AWR_TEST_BACKEND", "gloo")          # "nccl" (= RCCL) on a box with one GPU per rank
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    torch.distributed.init_process_group(backend, rank=rank, world_size=world, **({"device_id": torch.device("cuda", dev)} if backend == "nccl" else {}))
    import awr_amd
    import awr_oracle as O
    from awr_amd.trainer import TrainEngine
    res = {}
    for mode, steps in (("same", 2), ("split", 1)):
        torch.manual_seed(1234 + rank)                   # different initial weights per rank: the broadcast must fix that
        net = awr_amd.get_deconv_net(18, 14, 2).cuda()
        eng = TrainEngine(net, 2, 128, 1.0, coord_weight=1.0, lr=1e-3, process_group=torch.distributed.group.WORLD, use_graph=False, autotune=False,
                          native_rccl=os.environ.get("AWR_TEST_NATIVE") == "1")      # 1: the library's own communicator (awr_dp_*) exchanges the buckets
        assert eng.dp and eng.world == world and len(eng.sync.buckets) > 1
        losses, per_step = [], []
        for s in range(steps):
            img, jt = O.synth_batch(2, 128, 14, seed=70 + s + (0 if mode == "same" else 10 * rank))
            l, _ = eng.step(img.cuda(), jt.cuda())
            losses.append(float(l[2]))
            per_step.append(net.flat_params()[:net.n_active].cpu())       # replicas must stay bitwise equal after EVERY step
        torch.cuda.synchronize()
        res[mode] = {"params": net.flat_params()[:net.n_active].cpu(), "buffers": net._barena.cpu(), "losses": losses, "per_step": per_step}
    # sharded evaluation (Trainer.test): rank r scores batches r, r + world, ... of the test loader, the error rows are gathered and
    # re-assembled in dataset order: every rank must report the single-process mpe
    from awr_amd.config import Config
    from awr_amd.trainer import SyntheticHands, Trainer

    class Cfg(Config):
        net, kernel_size, batch_size, num_workers, max_epoch, output_dir, load_model, exp_id, use_hipgraph, vis_freq = \
            "resnet_18", 1.0, 4, 0, 1, out + "_work%d" % rank, "", "dp", False, 0
    torch.manual_seed(99)
    tr = Trainer(Cfg(), None, SyntheticHands(10, seed=2), process_group=torch.distributed.group.WORLD)
    res["test_mpe"] = float(tr.test(1))
    res["test_params"] = tr.net.flat_params()[:tr.net.n_active].cpu()
    torch.save(res, "%s.rank%d" % (out, rank))
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
