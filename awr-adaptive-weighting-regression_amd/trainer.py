"""The optimisation / evaluation step of the reference (train.py:107-131, :182-192; test.py:67-86)
as one fused, graph-capturable sequence of HIP kernels, plus single-node data parallelism.

`TrainEngine.step(img, jt_uvd_gt)` performs exactly the work of one reference iteration:

    offset_gt   = FM.joint2offset(jt_uvd_gt, img, ks, F)            (never materialised: fused into the loss)
    offset_pred = net(img)                                          (training-mode BN, running stats updated)
    jt_uvd_pred = FM.offset2joint_softmax(offset_pred, img, ks)
    loss        = coord_w * crit(jt_uvd_pred, jt_uvd_gt) + dense_w * crit(offset_pred, offset_gt)
    optimizer.zero_grad(); loss.backward(); optimizer.step()        (Adam/SGD with torch semantics)

with no host synchronisation: losses and predictions stay on the device until the caller reads
them.  Hourglass quirk (train.py:116-121): the reference runs the network `stacks` times per
iteration and keeps only the last stack's loss; one forward with the BN momentum compounded
`stacks` times is numerically identical and is what runs here.

Data parallel (new w.r.t. the reference, which is single-GPU): one process per GPU, each rank steps
its own shard of the minibatch, gradients are summed with RCCL all-reduce (torch.distributed backend
"nccl" == RCCL over xGMI) on the flat gradient arena and scaled by 1/world inside the optimiser
kernel.  BatchNorm statistics stay rank-local (what stock DDP does).
"""
import torch

from . import _lib as L

HUBER_DELTA = 0.01


class GradSync:
    """Data-parallel plumbing over flat arenas: parameter/buffer broadcast from rank 0 and a bucketed SUM
    all-reduce of the gradient arena [0, n).  Device-agnostic (RCCL on the GPUs, gloo in the CPU tests); the
    1/world averaging is folded into the optimiser kernel (`grad_scale`)."""

    def __init__(self, n, process_group=None, n_buckets=4, min_bucket=1 << 20):
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        nb = max(1, min(n_buckets, n // min_bucket or 1))
        edges = [round(i * n / nb / 4) * 4 for i in range(nb)] + [n]
        self.buckets = [(edges[i], edges[i + 1]) for i in range(nb) if edges[i + 1] > edges[i]]
        self.grad_scale = 1.0 / self.world

    def broadcast(self, *tensors):
        if self.world > 1:
            for t in tensors:
                torch.distributed.broadcast(t, 0, group=self.pg)

    def allreduce(self, flat_grad):
        if self.world > 1:
            for lo, hi in self.buckets:
                torch.distributed.all_reduce(flat_grad[lo:hi], group=self.pg)


class TrainEngine:
    def __init__(self, net, batch_size, img_size, kernel_size, coord_weight=0.0, dense_weight=1.0, lr=1e-3, weight_decay=0.0,
                 optimizer="adam", momentum=0.9, process_group=None, use_graph=True, n_buckets=4):
        if not next(net.parameters()).is_cuda:
            raise L.AwrError("TrainEngine needs the network on the GPU")
        self.net, self.B, self.H = net, batch_size, img_size
        self.ks, self.cw, self.dw = float(kernel_size), float(coord_weight), float(dense_weight)
        self.lr, self.wd, self.opt, self.momentum = float(lr), float(weight_decay), optimizer, float(momentum)
        self.J = net.J
        self.F = img_size // 2
        dev = net.device
        self.stage = net.nstage - 1
        net.train()
        self.plan = net.get_plan(batch_size, img_size, True, supervised=(self.stage,), bn_repeat=net.nstage)
        self.jt_gt = torch.zeros(batch_size, self.J, 3, device=dev)
        self.jt_pred = torch.zeros(batch_size, self.J, 3, device=dev)
        self.stat = torch.zeros(batch_size, self.J, 2, device=dev)
        self.g_jt = torch.zeros(batch_size, self.J, 3, device=dev)
        self.acc = torch.zeros(2, device=dev, dtype=torch.float64)
        self.losses = torch.zeros(3, device=dev)          # [coord, dense, total]
        n = net.n_active
        self.m = torch.zeros(n, device=dev)
        self.v = torch.zeros(n, device=dev) if optimizer == "adam" else None
        self.step_count = 0
        self.use_graph = use_graph
        self.graph = None
        self._warm = 0
        self.sync = GradSync(n, process_group, n_buckets)
        self.world = self.sync.world
        if self.world > 1:      # identical initial parameters and BN buffers on every rank
            self.sync.broadcast(net.flat_params(), net._barena)
            net.weights_changed()

    # ---- the captured part: repack -> forward -> head + losses -> backward -------------------------------
    def _core(self):
        net, plan = self.net, self.plan
        B, J, F, H = self.B, self.J, self.F, self.H
        s = L.stream()
        plan.refresh_weights()
        plan._run(plan.fwd_ops)
        out = plan.outputs[self.stage]
        gout = plan.grad_outs[self.stage]
        img = plan.img
        L.call("awr_head_forward", L.ptr(out), L.ptr(img), B, J, F, H, self.ks, L.ptr(self.jt_pred), L.ptr(self.stat), s)
        L.call("awr_zero_f64", L.ptr(self.acc), 2, s)
        L.call("awr_dense_loss", L.ptr(out), L.ptr(self.jt_gt), L.ptr(img), B, J, F, H, self.ks, HUBER_DELTA, self.dw,
               self.acc.data_ptr() + 8, L.ptr(gout), 0, s)
        need_coord_grad = self.cw != 0.0
        L.call("awr_huber", L.ptr(self.jt_pred), L.ptr(self.jt_gt), B * J * 3, HUBER_DELTA, self.cw, L.ptr(self.acc),
               L.ptr(self.g_jt) if need_coord_grad else None, 0, s)
        if need_coord_grad:
            L.call("awr_head_backward", L.ptr(out), L.ptr(img), L.ptr(self.jt_pred), L.ptr(self.stat), L.ptr(self.g_jt), B, J, F, H, self.ks,
                   L.ptr(gout), 1, s)
        L.call("awr_loss_finalize", L.ptr(self.acc), 2, L.ptr(self.losses), s)
        plan._run(plan.bwd_ops)

    def _optimizer(self):
        net = self.net
        n = net.n_active
        s = L.stream()
        scale = self.sync.grad_scale
        if self.opt == "adam":
            L.call("awr_adam_step", L.ptr(net.flat_params()), L.ptr(net.flat_grads()), L.ptr(self.m), L.ptr(self.v), n, self.lr, 0.9, 0.999,
                   1e-8, self.wd, self.step_count, scale, s)
        else:
            L.call("awr_sgd_step", L.ptr(net.flat_params()), L.ptr(net.flat_grads()), L.ptr(self.m), n, self.lr, self.momentum, self.wd,
                   self.step_count, scale, s)

    def _allreduce(self):
        self.sync.allreduce(self.net.flat_grads()[:self.net.n_active])

    def step(self, img, jt_uvd_gt):
        """One optimisation step on this rank's shard.  Returns (losses[coord,dense,total], jt_uvd_pred)
        as device tensors that are valid until the next step; nothing is synchronised."""
        plan = self.plan
        plan.img.copy_(img, non_blocking=True)
        self.jt_gt.copy_(jt_uvd_gt, non_blocking=True)
        if self.use_graph and self.graph is None and self._warm >= 2:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._core()
        if self.graph is not None:
            self.graph.replay()
        else:
            self._core()
            self._warm += 1
        for bn in plan.bns:
            bn.counter += plan.bn_repeat
        if self.world > 1:
            self._allreduce()
        self.step_count += 1
        self._optimizer()
        self.net.weights_changed()
        return self.losses, self.jt_pred

    def set_lr(self, lr):
        self.lr = float(lr)

    # ---- optimizer state in torch.optim layout (checkpoint compatibility, train.py:165-170) ----------------
    def optimizer_state_dict(self):
        net = self.net
        names = [k for k, _, kind in net._layout if kind in ("conv_w", "deconv_w", "conv_b", "bn_w", "bn_b")]
        state = {}
        for i, k in enumerate(names):
            if k in net._unused:
                continue
            o, n, s = net._poff[k]
            ent = {"step": torch.tensor(float(self.step_count))}
            if self.opt == "adam":
                ent["exp_avg"] = self.m[o:o + n].view(s).clone()
                ent["exp_avg_sq"] = self.v[o:o + n].view(s).clone()
            else:
                ent["momentum_buffer"] = self.m[o:o + n].view(s).clone()
            state[i] = ent
        group = {"lr": self.lr, "weight_decay": self.wd, "params": list(range(len(names)))}
        if self.opt == "adam":
            group.update(betas=(0.9, 0.999), eps=1e-8, amsgrad=False)
        else:
            group.update(momentum=self.momentum, dampening=0, nesterov=False)
        return {"state": state if self.step_count else {}, "param_groups": [group]}


class InferEngine:
    """test.py:67-86 without the per-sample host loop: img -> dense map -> joints, eval-mode BN."""

    def __init__(self, net, batch_size, img_size, kernel_size, use_graph=True):
        self.net, self.B, self.H, self.ks = net, batch_size, img_size, float(kernel_size)
        net.eval()
        self.plan = net.get_plan(batch_size, img_size, False)
        self.J, self.F = net.J, img_size // 2
        self.jt = torch.zeros(batch_size, self.J, 3, device=net.device)
        self.stage = net.nstage - 1
        self.use_graph, self.graph, self._warm = use_graph, None, 0

    def _core(self):
        plan = self.plan
        plan._run(plan.fwd_ops)
        L.call("awr_head_forward", L.ptr(plan.outputs[self.stage]), L.ptr(plan.img), self.B, self.J, self.F, self.H, self.ks, L.ptr(self.jt),
               None, L.stream())

    def __call__(self, img):
        self.net.sync_weights(self.plan)
        self.plan.img.copy_(img, non_blocking=True)
        if self.use_graph and self.graph is None and self._warm >= 2:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._core()
        if self.graph is not None:
            self.graph.replay()
        else:
            self._core()
            self._warm += 1
        return self.jt


def smoke_step(dev):
    """Tiny end-to-end check used by __graft_entry__.smoke(): one ResNet18-deconv train step (B=2)
    on the HIP path against the oracle."""
    import awr_oracle as O          # test infrastructure; only reachable from smoke()
    from .nets import ResNet18Deconv
    img, jt = O.synth_batch(2, 128, 14, seed=7)
    man = O.manifest_for("resnet_18", 14)
    sd = O.procedural_state(man, seed=1)
    net = ResNet18Deconv(14)
    net.load_state_dict(sd)
    net = net.cuda()
    eng = TrainEngine(net, 2, 128, 1.0, coord_weight=1.0, dense_weight=1.0, use_graph=False)
    losses, jt_pred = eng.step(img.to(dev), jt.to(dev))
    ost = {"step": 0, "m": {}, "v": {}}
    lo, lc, ld, grads, jt_o = O.train_step("resnet_18", sd, ost, img, jt, 1.0, 1.0, 1.0)
    got = float(losses[2])
    assert abs(got - float(lo)) <= 2e-4 * max(1e-3, abs(float(lo))), ("loss", got, float(lo))
    assert float((jt_pred.cpu() - jt_o).abs().max()) < 1e-4, "joints"
    w = net.state_dict()["layer1.0.conv1.weight"].cpu()
    d = (w - sd["layer1.0.conv1.weight"]).abs().flatten()
    assert float(torch.quantile(d, 0.9)) < 1e-4 and float(d.max()) < 2.1e-3, "params after Adam"
