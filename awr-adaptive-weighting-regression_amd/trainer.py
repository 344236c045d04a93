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
import numpy as np
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
        if self.pg is not None:
            for t in tensors:
                torch.distributed.broadcast(t, 0, group=self.pg)

    def allreduce(self, flat_grad):
        if self.world > 1:
            for lo, hi in self.buckets:
                torch.distributed.all_reduce(flat_grad[lo:hi], group=self.pg)

    def global_mean(self, total, count, device=None):
        """sum(total over ranks) / sum(count over ranks): every rank gets the SAME epoch metric (what drives the LR
        scheduler and the log lines), whatever its own shard looked like."""
        if self.world > 1:
            t = torch.tensor([float(total), float(count)], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t, group=self.pg)
            total, count = float(t[0]), float(t[1])
        return total / count if count else float("nan")


class TrainEngine:
    def __init__(self, net, batch_size, img_size, kernel_size, coord_weight=0.0, dense_weight=1.0, lr=1e-3, weight_decay=0.0,
                 optimizer="adam", momentum=0.9, process_group=None, use_graph=False, n_buckets=4, autotune=True, wgrad_streams=2,
                 nhwc_boundary=None, trace_buckets=False, native_rccl=None, accum="auto", winograd=None, _share=None):
        """accum: "auto" (default) | "ordered" | "blocked" | None (the process-wide mode, awr_amd.set_gemm_accum) -- accumulation order of the
        forward / data-gradient GEMMs of this engine's plan.  "blocked" is the parity mode (a conv's rounding error at torch-CPU's level, a few %
        slower); "auto" blocks only the launches with a long K extent (include/awr_hip.h: awr_set_gemm_accum), where an ordered chain's error
        is largest and blocking is cheapest; "ordered" is one chain per output element everywhere.
        winograd: None (the process-wide mode, awr_amd.set_conv_winograd) | False | True ("forward") | "full" -- Winograd F(2x2, 3x3) forward, or forward +
        data gradient + weight gradient, of the eligible stride-1 3x3 convolutions (include/awr_hip.h)."""
        if not next(net.parameters()).is_cuda:
            raise L.AwrError("TrainEngine needs the network on the GPU")
        self.net, self.B, self.H = net, batch_size, img_size
        self.ks, self.cw, self.dw = float(kernel_size), float(coord_weight), float(dense_weight)
        self.lr, self.wd, self.opt, self.momentum = float(lr), float(weight_decay), optimizer, float(momentum)
        self.J = net.J
        self.F = img_size // getattr(net, "downsample", 2)      # train.py:110: ft_sz = img_size / downsample
        dev = net.device
        self.stage = net.nstage - 1
        net.train()
        world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        import os
        self.dp = world > 1 or (process_group is not None and os.environ.get("AWR_FORCE_DP") == "1")   # test hook: 1-rank group
        # (single GPU: scattering the packed weight gradients bucket by bucket during the backward, like the data-parallel plans do, instead
        # of in one launch at the tail of the step was measured slower: 14.28-14.32 vs 14.00-14.05 ms, profiles/r03_summary.md)
        self.plan = net.get_plan(batch_size, img_size, True, supervised=(self.stage,), bn_repeat=net.nstage,
                                 n_buckets=n_buckets if self.dp else 1, accum=accum, winograd=winograd)
        self._accum, self._winograd = accum, winograd
        # weight-gradient GEMMs on extra HIP stream(s): co-resident DIFFERENT kernels fill each other's bubbles (0 = off); data
        # parallel: one more stream that finished gradient buckets (scatter + all-reduce) are handed to
        self.plan.set_streams(wgrad_streams, comm=(wgrad_streams > 0 and self.dp))
        # head + losses on the backbone's own NHWC layout (no transposes at the boundary, the dense map read once per step when
        # coord_weight == 0); AWR_NCHW_BOUNDARY=1 / nhwc_boundary=False keeps the reference-layout kernels (same-box A/B)
        import os as _os
        want_nhwc = (_os.environ.get("AWR_NCHW_BOUNDARY") != "1") if nhwc_boundary is None else bool(nhwc_boundary)
        self.nhwc = want_nhwc and self.plan.set_nhwc_boundary(True)
        if self.nhwc:
            self._pred, self._gpred, self._cp = self.plan.head_nhwc(self.stage)
            self._scratch = torch.zeros(int(L.lib.awr_head_nhwc_scratch(batch_size, self.J, self.F)), device=dev)
        self._autotune = bool(autotune)
        self._compiled = False
        self.jt_gt = torch.zeros(batch_size, self.J, 3, device=dev)
        self.jt_pred = torch.zeros(batch_size, self.J, 3, device=dev)
        self.stat = torch.zeros(batch_size, self.J, 2, device=dev)
        self.g_jt = torch.zeros(batch_size, self.J, 3, device=dev)
        self.acc = torch.zeros(2, device=dev, dtype=torch.float64)
        self.losses = torch.zeros(3, device=dev)          # [coord, dense, total]
        n = net.n_active
        self._children = {}
        if _share is not None:          # ragged-batch child: optimiser state lives in the parent engine
            self.m, self.v = _share.m, _share.v
        else:
            self.m = torch.zeros(n, device=dev)
            self.v = torch.zeros(n, device=dev) if optimizer == "adam" else None
        self.step_count = 0
        self.use_graph = use_graph
        self.graph = None
        self.sync = GradSync(n, process_group, n_buckets)
        self._n_buckets = n_buckets
        self.world = self.sync.world
        self._works, self.dpcomm = [], None
        self.trace_buckets, self._trace, self._tev = bool(trace_buckets), [], None
        if self.dp:             # identical initial parameters and BN buffers on every rank
            if _share is None:
                self.sync.broadcast(net.flat_params(), net._barena)
                net.weights_changed()
            self.use_graph = False          # the bucket markers interleave RCCL calls with the backward: run it eagerly
            # native_rccl (opt-in; AWR_NATIVE_RCCL=1): the library's own RCCL communicator (awr_dp_*, csrc/awr_dp.hip) exchanges the buckets
            # from inside the native replay -- no Python callback, no torch work objects; bootstrapped over `process_group`
            import os as _os2
            want_native = (_os2.environ.get("AWR_NATIVE_RCCL") == "1") if native_rccl is None else bool(native_rccl)
            if want_native and not trace_buckets:
                from .engine import DpComm
                self.dpcomm = _share.dpcomm if (_share is not None and getattr(_share, "dpcomm", None) is not None) else DpComm.from_process_group(process_group)
                self.plan.set_dp(self.dpcomm)
            g = net.flat_grads()
            # torch's NCCL work stream waits for the compute stream at the point of the call and runs concurrently with
            # whatever is enqueued afterwards: the rest of the backward overlaps the bucket's all-reduce over xGMI

            def hook(lo, hi):
                if self.trace_buckets:       # timestamps on the stream the bucket was handed to (the plan's comm stream)
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record()
                w = torch.distributed.all_reduce(g[lo:hi], group=process_group, async_op=True)
                self._works.append(w)
                if self.trace_buckets:       # (tracing serialises this stream behind the collective; untraced steps do not wait here)
                    w.wait()
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    self._trace.append((lo, hi, e0, e1))
            if self.dpcomm is None:
                self.plan.bucket_hook = hook

    # ---- the captured part: repack -> forward -> head + losses -> backward -------------------------------
    def _core(self):
        net, plan = self.net, self.plan
        B, J, F, H = self.B, self.J, self.F, self.H
        s = L.stream()
        plan.refresh_weights()
        plan.run_forward()
        if self.nhwc and not plan.nhwc:
            plan.set_nhwc_boundary(True)      # (the drop-in module shares plans and switches them back to the NCHW boundary)
        if self.nhwc:      # train.py:118-127 in one call on the NHWC map: joints, both Huber losses, d(loss)/d(map) written where the backward reads it
            # (self.acc starts at zero and awr_loss_finalize_reset leaves it zeroed: no fill launch per step)
            L.call("awr_head_loss_step_nhwc", self._pred, self._cp, L.ptr(plan.img), L.ptr(self.jt_gt), B, J, F, H, self.ks, HUBER_DELTA, self.cw, self.dw,
                   L.ptr(self._scratch), L.ptr(self.jt_pred), L.ptr(self.stat), L.ptr(self.g_jt), L.ptr(self.acc), self._gpred, s)
            L.call("awr_loss_finalize_reset", L.ptr(self.acc), 2, L.ptr(self.losses), s)
            plan.run_backward()
            return
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
        plan.run_backward()

    def bucket_timeline(self):
        """trace_buckets=True: where the gradient exchange of the LAST step sat relative to its backward -- milliseconds from the start
        of the step (stream timestamps): {"backward_end_ms", "buckets": [{"lo", "hi", "mbytes", "start_ms", "end_ms"}], "tail_ms"}.
        backward_end_ms is taken after the end-of-backward join (all side streams and the last bucket's exchange are in); tail_ms = the
        time from handing the LAST bucket over to that join -- the only exchange nothing is left to overlap with; every earlier bucket
        whose end_ms lies before the last bucket's start_ms ran entirely under the backward."""
        if not self.trace_buckets or self._tev is None:
            return None
        torch.cuda.synchronize()
        t0 = self._tev[0]
        bk = [{"lo": lo, "hi": hi, "mbytes": round((hi - lo) * 4e-6, 2), "start_ms": round(t0.elapsed_time(e0), 3), "end_ms": round(t0.elapsed_time(e1), 3)}
              for lo, hi, e0, e1 in self._trace]
        end = round(t0.elapsed_time(self._tev[1]), 3)
        return {"backward_end_ms": end, "buckets": bk, "tail_ms": round(end - bk[-1]["start_ms"], 3) if bk else 0.0}

    def dense_map(self, stage=None):
        """(B, 4J, F, F) dense map of the last step in the reference's layout."""
        return self.plan.dense_map(self.stage if stage is None else stage)

    def timed_core(self, every=False):
        """The captured part once more, serially, with a HIP-event pair around every launch (bench.py's roofline):
        -> {launch name: seconds} of the conv / stem launches (`every`: of all launches).  Not an optimisation step (no optimiser update)."""
        self.plan.refresh_weights()
        per = self.plan.timed("fwd", every)
        per.update(self.plan.timed("bwd", every))
        return per

    def timed_hbm(self, reps=5):
        """Serial passes with an event pair around the HBM-bound launches of the step that live outside the plan (bench.py's
        `roofline_hbm`): the head + loss part (train.py:118-127) and the optimiser (on scratch copies: no parameter moves).
        -> {name: seconds per call}.  Call after a step (the buffers hold real data)."""
        B, J, F, H = self.B, self.J, self.F, self.H
        s = L.stream()
        plan = self.plan
        res = {}

        def t(name, fn):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) * 1e-3 / reps
        if self.nhwc:
            def step_nhwc():
                L.call("awr_head_loss_step_nhwc", self._pred, self._cp, L.ptr(plan.img), L.ptr(self.jt_gt), B, J, F, H, self.ks, HUBER_DELTA, self.cw, self.dw,
                       L.ptr(self._scratch), L.ptr(self.jt_pred), L.ptr(self.stat), L.ptr(self.g_jt), L.ptr(self.acc), self._gpred, s)
            t("head_loss_step_nhwc", step_nhwc)
            t("head_forward_nhwc", lambda: L.call("awr_head_forward_nhwc", self._pred, self._cp, L.ptr(plan.img), B, J, F, H, self.ks, L.ptr(self._scratch),
                                                  L.ptr(self.jt_pred), L.ptr(self.stat), s))
        else:
            out, gout, img = plan.outputs[self.stage], plan.grad_outs[self.stage], plan.img
            t("head_forward", lambda: L.call("awr_head_forward", L.ptr(out), L.ptr(img), B, J, F, H, self.ks, L.ptr(self.jt_pred), L.ptr(self.stat), s))
            t("dense_loss", lambda: L.call("awr_dense_loss", L.ptr(out), L.ptr(self.jt_gt), L.ptr(img), B, J, F, H, self.ks, HUBER_DELTA, self.dw,
                                           self.acc.data_ptr() + 8, L.ptr(gout), 0, s))
            t("head_backward", lambda: L.call("awr_head_backward", L.ptr(out), L.ptr(img), L.ptr(self.jt_pred), L.ptr(self.stat), L.ptr(self.g_jt), B, J, F, H,
                                              self.ks, L.ptr(gout), 1, s))
        if self.opt == "adam":
            n = self.net.n_active
            p, g, m, v = (x[:n].clone() for x in (self.net.flat_params(), self.net.flat_grads(), self.m, self.v))
            t("adam_step", lambda: L.call("awr_adam_step", L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), n, self.lr, 0.9, 0.999, 1e-8, self.wd, max(self.step_count, 1),
                                          self.sync.grad_scale, s))
        self.acc.zero_()          # the passes above accumulated into the loss accumulators
        return res

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

    def _ragged(self, b):
        """Engine for a smaller (last-of-epoch) batch: its own static plan, the SAME optimiser state and step counter
        (the reference's DataLoader keeps the ragged last batch, train.py:109 drop_last=False)."""
        eng = self._children.get(b)
        if eng is None:
            eng = TrainEngine(self.net, b, self.H, self.ks, self.cw, self.dw, self.lr, self.wd, self.opt, self.momentum,
                              process_group=self.sync.pg, use_graph=False, n_buckets=self._n_buckets, autotune=False,
                              wgrad_streams=0, accum=self._accum, winograd=self._winograd, _share=self)
            self._children[b] = eng
        eng.lr, eng.step_count = self.lr, self.step_count
        return eng

    def step(self, img, jt_uvd_gt):
        """One optimisation step on this rank's shard.  Returns (losses[coord,dense,total], jt_uvd_pred)
        as device tensors that are valid until the next step; nothing is synchronised."""
        if img.shape[0] != self.B:
            if not 0 < img.shape[0] < self.B:
                raise L.AwrError("TrainEngine built for batches of %d got %d images" % (self.B, img.shape[0]))
            eng = self._ragged(int(img.shape[0]))
            out = eng.step(img, jt_uvd_gt)
            self.step_count = eng.step_count
            return out
        plan = self.plan
        plan.img.copy_(img, non_blocking=True)
        self.jt_gt.copy_(jt_uvd_gt, non_blocking=True)
        if not self._compiled:
            self.compile()
        if self.trace_buckets:
            del self._trace[:]
            self._tev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            self._tev[0].record()
        try:
            if self.graph is not None:
                self.graph.replay()
            else:
                self._core()
        except Exception:
            # the NHWC loss call ADDS onto self.acc and only awr_loss_finalize_reset re-arms it: a step that dies between the two would
            # leave the next step's losses inflated
            self.acc.zero_()
            raise
        if self.trace_buckets:
            self._tev[1].record()
        self.net._counters += plan.bn_repeat          # num_batches_tracked of every BatchNorm (host side)
        if self.dp:
            for w in self._works:           # compute stream waits for the bucket all-reduces before the optimiser
                w.wait()
            self._works.clear()
        self.step_count += 1
        self._optimizer()
        self.net.weights_changed()
        return self.losses, self.jt_pred

    def compile(self, img=None, jt_uvd_gt=None):
        """One-off set-up of the static plan, NOT an optimisation step: runs the captured part once eagerly on the batch in the
        input buffers (kernel warm-up; its only side effect -- the BatchNorm running statistics -- is rolled back), times the GEMM
        tile candidates of every launch in place (engine.Plan.autotune) and captures repack + forward + head + losses + backward,
        with the weight-gradient side streams forked and joined inside the capture, as ONE hipGraph.  `step()` calls it on its
        first use; call it directly (optionally with a representative batch) to keep that cost out of the first step."""
        if self._compiled:
            return
        if img is not None:
            self.plan.img.copy_(img, non_blocking=True)
            self.jt_gt.copy_(jt_uvd_gt, non_blocking=True)
        self._compiled = True
        tune = self._autotune and not self.plan.tuned and not self.plan.det
        if not (self.use_graph or tune):
            return
        hook, self.plan.bucket_hook = self.plan.bucket_hook, None          # no collectives during set-up
        dpc = getattr(self, "dpcomm", None)
        if dpc is not None:
            self.plan.set_dp(None)
        keep = self.net._barena.clone()
        self._core()
        if tune:
            self.plan.autotune(cache_key="train/%s/J%d/B%d/H%d" % (type(self.net).__name__ + str(self.net.nstage), self.J, self.B, self.H))
        if self.use_graph:
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._core()
        self.net._barena.copy_(keep)
        self.plan.bucket_hook = hook
        if dpc is not None:
            self.plan.set_dp(dpc)

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


def _load_opt_into(engine, sd):
    """Inverse of optimizer_state_dict(): accepts a stock torch.optim.Adam/SGD state dict (train.py:84)."""
    net = engine.net
    names = [k for k, _, kind in net._layout if kind in ("conv_w", "deconv_w", "conv_b", "bn_w", "bn_b")]
    steps = []
    for i, k in enumerate(names):
        ent = sd.get("state", {}).get(i)
        if ent is None or k in net._unused:
            continue
        o, n, s = net._poff[k]
        if engine.opt == "adam":
            engine.m[o:o + n].copy_(ent["exp_avg"].reshape(-1))
            engine.v[o:o + n].copy_(ent["exp_avg_sq"].reshape(-1))
            steps.append(int(float(ent["step"])))
        else:
            engine.m[o:o + n].copy_(ent["momentum_buffer"].reshape(-1))
    if steps:
        engine.step_count = max(steps)
    elif engine.opt != "adam" and sd.get("state"):
        engine.step_count = max(engine.step_count, 1)


TrainEngine.load_optimizer_state_dict = lambda self, sd: _load_opt_into(self, sd)


class InferEngine:
    """test.py:67-86 without the per-sample host loop: img -> dense map -> joints, eval-mode BN."""

    def __init__(self, net, batch_size, img_size, kernel_size, use_graph=False, autotune=True, parity=False, winograd=None):
        """parity=True: blocked accumulation in the GEMMs of this engine's plan (awr_amd.set_gemm_accum) -- scoring passes (test.py:67-86) care
        about the last digits of the joints, not about the last few per cent of throughput.
        winograd: None (the process-wide mode, awr_amd.set_conv_winograd) | False | True -- the eligible stride-1 3x3 convolutions of the eval plan as
        Winograd F(2x2, 3x3) with the folded BatchNorm / residual add in its epilogue (Hourglass: instead of the fused conv2 + conv3 launch)."""
        self.net, self.B, self.H, self.ks = net, batch_size, img_size, float(kernel_size)
        self.parity = bool(parity)
        if self.parity and (int(L.lib.awr_get_gemm_products()) != 1 or int(L.lib.awr_get_gemm_staging()) == 0):
            # the blocked kernel exists for FP32-MFMA + LDS-DMA staging only: a scoring pass in the split-operand mode (config.gemm_products = 6)
            # or with register staging runs ordered and says so, instead of failing at its first launch
            import warnings
            warnings.warn("InferEngine(parity=True): blocked accumulation is not available with gemm_products = %d / staging = %d; "
                          "scoring with ordered accumulation" % (L.lib.awr_get_gemm_products(), L.lib.awr_get_gemm_staging()))
            self.parity = False
        net.eval()
        self.plan = net.get_plan(batch_size, img_size, False, accum="blocked" if self.parity else None, winograd=winograd)
        if self.plan.n_side == 0:          # forward branches (ResNet downsample projections, Hourglass skip residuals) run beside the main chain
            self.plan.set_streams(4)
        self._autotune, self._compiled = bool(autotune), False
        self.J, self.F = net.J, img_size // getattr(net, "downsample", 2)
        self.jt = torch.zeros(batch_size, self.J, 3, device=net.device)
        self.stage = net.nstage - 1
        self.use_graph, self.graph = use_graph, None
        import os as _os
        self.nhwc = _os.environ.get("AWR_NCHW_BOUNDARY") != "1" and self.plan.set_nhwc_boundary(True)
        if self.nhwc:
            self._pred, _, self._cp = self.plan.head_nhwc(self.stage)
            self._scratch = torch.zeros(int(L.lib.awr_head_nhwc_scratch(batch_size, self.J, self.F)), device=net.device)

    def _core(self):
        plan = self.plan
        if self.nhwc and not plan.nhwc:
            plan.set_nhwc_boundary(True)
        plan.run_forward()
        if self.nhwc:
            L.call("awr_head_forward_nhwc", self._pred, self._cp, L.ptr(plan.img), self.B, self.J, self.F, self.H, self.ks, L.ptr(self._scratch),
                   L.ptr(self.jt), None, L.stream())
            return
        L.call("awr_head_forward", L.ptr(plan.outputs[self.stage]), L.ptr(plan.img), self.B, self.J, self.F, self.H, self.ks, L.ptr(self.jt),
               None, L.stream())

    def __call__(self, img):
        self.net.sync_weights(self.plan)
        self.plan.img.copy_(img, non_blocking=True)
        if not self._compiled:           # one-off: eager warm-up run, GEMM tile autotune, hipGraph capture
            self._compiled = True
            tune = self._autotune and not self.plan.tuned and not self.plan.det
            if self.use_graph or tune:
                self._core()
                if tune:
                    self.plan.autotune(cache_key="infer/%s/J%d/B%d/H%d" % (type(self.net).__name__ + str(self.net.nstage), self.J, self.B, self.H))
                if self.use_graph:
                    torch.cuda.synchronize()
                    self.graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.graph):
                        self._core()
        if self.graph is not None:
            self.graph.replay()
        else:
            self._core()
        return self.jt


class _Plateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, 'min', patience=2, min_lr=1e-8) (train.py:90),
    defaults factor=0.1, threshold=1e-4 'rel', cooldown=0 -- host-side, drives TrainEngine.set_lr."""

    def __init__(self, lr, patience=2, factor=0.1, min_lr=1e-8, threshold=1e-4):
        self.lr, self.patience, self.factor, self.min_lr, self.threshold = lr, patience, factor, min_lr, threshold
        self.best, self.bad = float("inf"), 0

    def step(self, metric):
        if metric < self.best * (1.0 - self.threshold):
            self.best, self.bad = metric, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            self.lr = max(self.lr * self.factor, self.min_lr)
            self.bad = 0
        return self.lr


class SyntheticHands(torch.utils.data.Dataset):
    """Stand-in for dataloader/nyu_loader.py (the NYU files are not redistributable): yields the same 6-tuple
    `(img[1,S,S], jt_xyz[J,3], jt_uvd[J,3], center_xyz[3], M[3,3], cube[3])` (nyu_loader.py:66) with joints that lie
    on a synthetic hand surface, so a network can actually fit them."""

    def __init__(self, n, img_size=128, jt_num=14, seed=0):
        g = torch.Generator().manual_seed(seed)
        S = img_size
        yy, xx = torch.meshgrid(torch.arange(S).float(), torch.arange(S).float(), indexing="ij")
        c = S / 2 + (torch.rand(n, 2, generator=g) * 2 - 1) * (S / 16)
        R = 0.31 * S
        r2 = (xx - c[:, 0].view(n, 1, 1)) ** 2 + (yy - c[:, 1].view(n, 1, 1)) ** 2
        depth = (0.5 * r2 / (R * R) - 0.3).clamp(-1.0, 0.98)
        self.img = torch.where(r2 < R * R, depth, torch.ones(())).view(n, 1, S, S).contiguous()
        ang = torch.rand(n, jt_num, generator=g) * 6.2831853
        rad = torch.rand(n, jt_num, generator=g).sqrt() * 0.8 * R
        px = (c[:, 0:1] + rad * torch.cos(ang)).round().clamp(0, S - 1).long()
        py = (c[:, 1:2] + rad * torch.sin(ang)).round().clamp(0, S - 1).long()
        d = self.img[torch.arange(n).view(n, 1), 0, py, px]
        self.jt_uvd = torch.stack([(px.float() + 0.5) * 2 / S - 1, (py.float() + 0.5) * 2 / S - 1, d], -1)
        self.center = torch.tensor([0.0, 0.0, 750.0]).expand(n, 3).contiguous()
        self.cube = torch.tensor([300.0, 300.0, 300.0]).expand(n, 3).contiguous()
        s = 0.45                                   # crop matrix: original-image pixels -> crop pixels
        self.M = torch.tensor([[s, 0.0, 64.0 - 320.0 * s], [0.0, s, 64.0 - 240.0 * s], [0.0, 0.0, 1.0]]).expand(n, 3, 3).contiguous()
        # ground-truth xyz consistent with the evaluator's uvd -> xyz chain, normalised like loader.py:242-260
        from .evaluator import uvd2xyz
        uv = (self.jt_uvd[:, :, :2] + 1) * S / 2.0
        uv = (uv - torch.tensor([64.0 - 320.0 * s, 64.0 - 240.0 * s])) / s
        uvd = torch.cat([uv, self.jt_uvd[:, :, 2:] * 150.0 + 750.0], -1)
        xyz = torch.from_numpy(uvd2xyz(uvd.numpy(), (588.03, 587.07, 320.0, 240.0), -1))
        self.jt_xyz = (xyz - self.center.view(n, 1, 3)) / 150.0
        self.img_size, self.jt_num, self.paras, self.flip = S, jt_num, (588.03, 587.07, 320.0, 240.0), -1

    def __len__(self):
        return self.img.shape[0]

    def __getitem__(self, i):
        return self.img[i], self.jt_xyz[i], self.jt_uvd[i], self.center[i], self.M[i], self.cube[i]


class Trainer:
    """The reference's Trainer (train.py:27-227, test.py:20-110) on the MI355X engines: same config attributes, same
    log lines, same checkpoint files, with the per-sample host loops removed.  `train_data` / `test_data` are any
    map-style datasets yielding the NYU 6-tuple (the reference's own `NYU(...)` objects work unchanged)."""

    def __init__(self, config, train_data=None, test_data=None, process_group=None):
        import os
        from . import resnet_deconv, hourglass
        from .evaluator import EvalUtil
        self.config, self.EvalUtil = config, EvalUtil
        if train_data is None and test_data is None:      # train.py:58-61: datasets come from the config
            if config.dataset != "nyu":
                raise ValueError("only the NYU loader is provided (dataset=%r): pass train_data / test_data objects" % config.dataset)
            from .nyu_data import NYU
            root = os.path.join(config.data_dir, config.dataset)
            if not os.path.isdir(os.path.join(root, "test")):
                raise FileNotFoundError("NYU data not found under %s (expects train/ test/ center_*_refined.txt, dataloader/nyu_loader.py:38-49)" % root)
            kw = dict(img_size=config.img_size, cube=config.cube, jt_num=config.jt_num)
            self._render = {}
            if getattr(config, "device_loader", False):
                # frames decoded once into a uint16 memmap, resident in HBM for the run; datasets yield 200-byte parameter blocks and the
                # image batch is produced by one kernel launch per step (nyu_device.py)
                from . import nyu_device as DV
                NYU = DV.DeviceNYU
                for phase in ("train", "test"):
                    if os.path.isdir(os.path.join(root, phase)):
                        store = DV.FrameStore(DV.build_frame_cache(root, phase))
                        self._render[phase] = DV.Renderer(store, config.img_size, config.batch_size)
            if os.path.isdir(os.path.join(root, "train")):
                train_data = NYU(root, "train", aug_para=config.augment_para, **kw)
            test_data = NYU(root, "test", **kw)
        self.trainData, self.testData, self.pg = train_data, test_data, process_group
        self._render = getattr(self, "_render", {})
        self.rank = torch.distributed.get_rank(process_group) if process_group is not None else 0
        self.work_dir = os.path.join(config.output_dir, config.dataset, "checkpoint_" + config.exp_id)
        self.result_dir = os.path.join(self.work_dir, "results")
        os.makedirs(self.result_dir, exist_ok=True)
        self.log = open(os.path.join(self.work_dir, "%s_%s.log" % (config.net, config.log_id)), "a") if self.rank == 0 else None
        self._msg("-------------------start programming-------------------", stdout=False)
        for k in sorted(k for k in dir(config) if not k.startswith("_") and not callable(getattr(config, k))):     # train.py:44-46
            self._msg(str(k) + ":" + str(getattr(config, k)))
        if "resnet" in config.net:
            self.net = resnet_deconv.get_deconv_net(int(config.net.split("_")[1]), config.jt_num, config.downsample)
            self.stacks = 1
        else:
            self.stacks = int(config.net.split("_")[1])
            self._msg("hourglass stacks:{}".format(self.stacks))
            self.net = hourglass.PoseNet(config.net, config.jt_num)
        self.net = self.net.cuda()
        self.best_records = {"epoch": 0, "MPE": 1e10, "AUC": 0}
        try:
            from .vis_tool import VisualUtil
            self._vis = VisualUtil(config.dataset)                   # train.py:43
        except ValueError:
            self._vis = None
        if getattr(config, "gemm_products", 1) != 1:      # opt-in split-operand GEMMs (process-wide)
            from . import set_gemm_products
            set_gemm_products(config.gemm_products)
        self.engine = TrainEngine(self.net, config.batch_size, config.img_size, config.kernel_size, config.coord_weight, config.dense_weight,
                                  config.lr, config.weight_decay, config.optimizer, process_group=process_group,
                                  use_graph=getattr(config, "use_hipgraph", False), accum=getattr(config, "accum", "auto"),
                                  winograd=getattr(config, "winograd", None))
        if config.load_model and os.path.exists(config.load_model):
            self._msg("loading model from {}".format(config.load_model))
            pth = torch.load(config.load_model, map_location="cpu", weights_only=False)     # trusted project artefact (best_records may hold numpy scalars)
            self.net.load_state_dict(pth["model"])
            if "optimizer" in pth:
                self.engine.load_optimizer_state_dict(pth["optimizer"])
            if "best_records" in pth:
                self.best_records = pth["best_records"]
        # train.py:94-96: the learning rate is force-reset to config.lr after loading
        self.engine.set_lr(config.lr)
        self._plateau = _Plateau(config.lr) if config.scheduler == "auto" else None
        self._msg("learning rate: {:.1e}".format(config.lr))

    def _msg(self, msg, stdout=True):
        if self.rank != 0:
            return
        if stdout:
            print(msg)
        print(msg, file=self.log)

    def _images(self, first, phase):
        """First element of a batch -> (B, 1, S, S) float32 on the GPU: host-loader images are copied, device-loader parameter blocks
        (uint8, nyu_device.BLOCK_BYTES each) are rendered from the HBM-resident frames by one kernel launch."""
        if first.dtype == torch.uint8:
            r = self._render.get(phase)
            if r is None:
                raise L.AwrError("the dataset yields device-loader parameter blocks but no frame store was built for phase %r" % phase)
            return r(first)
        return first.cuda(non_blocking=True).float()

    def _loader(self, data, shuffle, epoch=0):
        """train.py:109: DataLoader(batch_size, shuffle=True, num_workers) -- drop_last=False like the reference: the ragged last
        batch of an epoch is trained on (TrainEngine keeps a second static plan for it).  Data parallel: every rank iterates its
        own 1/world shard of the epoch's permutation (DistributedSampler pads the shards to equal length, so all ranks see the
        same batch sizes and the 1/world gradient average stays the global mean), batch_size images per rank and step."""
        sampler = None
        if self.pg is not None:
            sampler = torch.utils.data.distributed.DistributedSampler(data, num_replicas=torch.distributed.get_world_size(self.pg), rank=self.rank,
                                                                      shuffle=shuffle, drop_last=False)
            sampler.set_epoch(epoch)
        return torch.utils.data.DataLoader(data, batch_size=self.config.batch_size, shuffle=shuffle and sampler is None, sampler=sampler,
                                           num_workers=int(getattr(self.config, "num_workers", 0)), drop_last=False)

    def train(self):
        cfg, eng = self.config, self.engine
        dev = self.net.device
        # train.py:102: ONE evaluator for the whole run -- the reference's `train mpe` (which drives ReduceLROnPlateau) is the
        # running mean over every frame fed since the start of training, not a per-epoch value.  Reproduced on purpose.
        ev = self.EvalUtil(self.trainData.img_size, self.trainData.paras, self.trainData.flip, self.trainData.jt_num)
        for epoch in range(self.best_records["epoch"] + 1, cfg.max_epoch + 1):
            self.net.train()
            lsum, lcnt, pend, last_mean = torch.zeros(3, device=dev), 0, [], float("nan")

            def drain():
                for jt, a, b, c, d in pend:
                    ev.feed_batch(jt.cpu().numpy(), a.numpy(), b.numpy(), c.numpy(), d.numpy())
                del pend[:]

            def meter():                # mean of the per-step losses since the last print, identical on every rank
                if self.pg is None or not lcnt:
                    return (lsum / max(lcnt, 1)).tolist()
                t = torch.cat([lsum.double(), torch.tensor([float(lcnt)], dtype=torch.float64, device=dev)])
                torch.distributed.all_reduce(t, group=self.pg)
                return (t[:3] / t[3]).tolist()
            for ii, (img, jt_xyz_gt, jt_uvd_gt, center_xyz, M, cube) in enumerate(self._loader(self.trainData, True, epoch)):
                losses, jt_pred = eng.step(self._images(img, "train"), jt_uvd_gt.cuda(non_blocking=True))
                lsum += losses                                      # device-side meter: no loss.item() per iteration
                lcnt += 1
                pend.append((jt_pred.clone(), jt_xyz_gt, center_xyz, M, cube))
                if (ii + 1) % cfg.print_freq == 0:
                    l = meter()
                    last_mean = l[2]
                    self._msg("[epoch: {:02d}][train loss: {:.5f}][offset_loss: {:.5f}][coord_loss: {:.5f}]".format(epoch, l[2], l[1], l[0]))
                    lsum.zero_()
                    lcnt = 0
                    drain()
            drain()
            # rank-uniform metric: every rank must feed the SAME number to its LR scheduler, or the replicas would step the
            # (identical, all-reduced) gradients with different learning rates and silently diverge
            if ev._err:
                e = np.concatenate(ev._err, 0).astype(np.float64)
                train_mpe = eng.sync.global_mean(e.mean(1).sum(), e.shape[0], device=dev)
            else:
                train_mpe = eng.sync.global_mean(0.0, 0, device=dev)    # an epoch without a single batch (empty shard)
            last = meter()[2] if lcnt else last_mean                     # meter is reset at every print (train.py:139)
            self._msg("[epoch {:02d}], [train loss {:.5f}], [train mpe {:.5f}], [lr {:.1e}]".format(epoch, last, train_mpe, eng.lr))
            if cfg.scheduler == "auto":
                if train_mpe == train_mpe:                               # NaN (nothing evaluated yet) never steps the plateau counter
                    eng.set_lr(self._plateau.step(train_mpe))
            elif cfg.scheduler == "step":                          # StepLR(step_size=cfg.step, gamma=0.1).step(epoch)
                eng.set_lr(cfg.lr * 0.1 ** (epoch // cfg.step))
            if self.testData is not None:
                self.test(epoch)
            if self.rank == 0:
                import os
                torch.save({"model": self.net.state_dict(), "optimizer": eng.optimizer_state_dict(), "best_records": self.best_records},
                           os.path.join(self.work_dir, "epoch_{}.pth".format(epoch)))
        if self.log:
            self.log.flush()

    @torch.no_grad()
    def test(self, epoch=0):
        """train.py:178-227 / test.py:51-110 -- DataLoader(batch_size, shuffle=False, num_workers) like the reference (worker processes
        decode the 8 252 NYU PNGs while the GPU runs), no per-sample host loop.  Data parallel: rank r evaluates batches r, r + world, ...
        of that loader's order; the per-frame error rows and original-image uvd predictions are gathered and re-assembled in dataset order
        on every rank, so mpe / AUC / the results txt are identical to the single-process run and rank-uniform."""
        import os
        import numpy as np
        cfg = self.config
        world = torch.distributed.get_world_size(self.pg) if self.pg is not None else 1
        # config.parity_infer = True scores with blocked accumulation (eval-mode plans measure no gain from it: off by default since round 6)
        inf = self._last_infer = InferEngine(self.net, cfg.batch_size, cfg.img_size, cfg.kernel_size, use_graph=False,
                                             parity=bool(getattr(cfg, "parity_infer", False)), winograd=getattr(cfg, "winograd", None))
        ev = self.EvalUtil(self.testData.img_size, self.testData.paras, self.testData.flip, self.testData.jt_num)
        n, bs = len(self.testData), cfg.batch_size
        mine = [b for b in range((n + bs - 1) // bs) if b % world == self.rank]
        idx = [i for b in mine for i in range(b * bs, min(n, (b + 1) * bs))]
        loader = torch.utils.data.DataLoader(torch.utils.data.Subset(self.testData, idx), batch_size=bs, shuffle=False,
                                             num_workers=int(getattr(cfg, "num_workers", 0)), drop_last=False)
        pad = None
        for k, (img, jt_xyz_gt, jt_uvd_gt, center_xyz, M, cube) in enumerate(loader):
            nb = img.shape[0]
            x = self._images(img, "test")
            if img.dtype == torch.uint8:                             # device loader: the host only ever saw parameter blocks
                img = x[:1].cpu() if (getattr(cfg, "vis_freq", 0) and (mine[k] + 1) % cfg.vis_freq == 0 and self._vis is not None) else None
            if nb < bs:                                              # ragged last batch: fill the static plan's batch with zeros, drop them after
                if pad is None:
                    pad = torch.zeros((bs,) + tuple(x.shape[1:]), device=x.device)
                pad.zero_()
                pad[:nb] = x
                x = pad
            jt = inf(x)[:nb].cpu().numpy()
            ev.feed_batch(jt, jt_xyz_gt.numpy(), center_xyz.numpy(), M.numpy(), cube.numpy())
            ib = mine[k] + 1
            if getattr(cfg, "vis_freq", 0) and ib % cfg.vis_freq == 0 and self._vis is not None:    # train.py:203-213
                half = cfg.img_size / 2.0
                self._vis.plot(img[0].numpy(), os.path.join(self.result_dir, "test_epoch_{}_iter_{}.png".format(epoch, ib)),
                               (jt[0] + 1) * half, (jt_uvd_gt[0].numpy() + 1) * half)
        self.net.train()
        if world > 1:          # every rank ends up with the whole test set, in dataset order
            J = self.testData.jt_num
            err = np.concatenate(ev._err, 0) if ev._err else np.zeros((0, J), np.float32)
            uvd = np.asarray(ev.jt_uvd_pred, np.float32).reshape(-1, J, 3)
            parts = [None] * world
            torch.distributed.all_gather_object(parts, (idx, err, uvd), group=self.pg)
            full_e, full_u = np.zeros((n, J), err.dtype), np.zeros((n, J, 3), np.float32)
            for ids, e, u in parts:
                full_e[ids], full_u[ids] = e, u
            ev._err, ev.jt_uvd_pred = [full_e], list(full_u)
        mpe, mid, auc, pck, thresh = ev.get_measures()
        if self.rank == 0:
            ev.plot_pck(os.path.join(self.work_dir, "test_pck_epoch_{}.png".format(epoch)), pck, thresh)                # train.py:216
        if epoch in (0, -1) and self.rank == 0:                      # train.py:217-221 / test.py:103-108
            jt_uvd = np.array(ev.jt_uvd_pred, dtype=np.float32)
            np.savetxt(os.path.join(self.work_dir, "test_%.3f.txt" % mpe), jt_uvd.reshape([jt_uvd.shape[0], cfg.jt_num * 3]), fmt="%.3f")
        self._msg("[epoch {:2d}], [test mpe {:.3f}], [lr {:.1e}]".format(epoch, mpe, self.engine.lr))
        return mpe
