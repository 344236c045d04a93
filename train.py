"""Entry point with the reference's usage (`python train.py`, settings in config.py -- reference train.py:231-236) on the
MI355X engines.

    python train.py                                   # NYU from config.data_dir, one GPU
    python train.py --set net=resnet_18 kernel_size=1 batch_size=64 --synthetic 4096
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train.py     # data parallel, RCCL

`--set key=value ...` overrides config entries for this run (values are parsed as Python literals when possible);
`--synthetic N` trains on N procedurally generated hands instead of NYU (no dataset on this machine);
`--test-only` runs the evaluation pass of test.py on the loaded checkpoint.
"""
import argparse
import ast
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def parse_overrides(items):
    out = {}
    for it in items or []:
        k, _, v = it.partition("=")
        try:
            out[k] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            out[k] = v
    return out


def main(argv=None, test_only=False):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--set", nargs="*", default=[], metavar="key=value")
    ap.add_argument("--synthetic", type=int, default=0, metavar="N")
    ap.add_argument("--test-only", action="store_true")
    args = ap.parse_args(argv)

    import torch
    import awr_amd
    from awr_amd.config import Config
    from awr_amd.trainer import SyntheticHands, Trainer

    cfg = Config(**parse_overrides(args.set))
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", str(cfg.gpu_id))))
    torch.cuda.set_device(local)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        pg = torch.distributed.group.WORLD
    data = (None, None)
    if args.synthetic:
        data = (SyntheticHands(args.synthetic, seed=1, img_size=cfg.img_size, jt_num=cfg.jt_num),
                SyntheticHands(max(cfg.batch_size, args.synthetic // 8), seed=2, img_size=cfg.img_size, jt_num=cfg.jt_num))
    trainer = Trainer(cfg, data[0], data[1], process_group=pg)
    if test_only or args.test_only:
        trainer.test(-1)
    else:
        trainer.train()
    if pg is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
