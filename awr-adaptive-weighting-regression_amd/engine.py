"""Binding of the network-level engine of libawr_hip.so (csrc/awr_net.hip; include/awr_hip.h "Network-level API").

The static execution plans of the AWR backbones -- buffer allocation, the forward launch list, the backward list derived by
a static autograd, gradient buckets, weight repacking, side-stream scheduling, GEMM tile autotuning -- are built and replayed
natively.  This module only wraps the handles: `NetHandle` (checkpoint layout + arenas bound to torch tensors) and `Plan`
(one (batch, size, mode) instance whose boundary tensors -- depth batch, NCHW dense maps, their gradients -- are torch tensors
so that the head / loss kernels and autograd can reach them).
"""
import ctypes as C
import json
import os

import torch

from . import _lib as L

PARAM_KINDS = ("conv_w", "deconv_w", "conv_b", "bn_w", "bn_b")
_KIND_NAMES = ("conv_w", "deconv_w", "conv_b", "bn_w", "bn_b", "bn_mean", "bn_var", "counter")


def plan_buckets(writes, n_active, n_buckets):
    """Reference statement of the native bucket planner (csrc/awr_net.hip: plan_buckets), kept for the CPU unit test.

    writes: list of (arena_lo, arena_hi, ready_op) -- the gradient occupying [lo,hi) floats is final once backward op number
    `ready_op` has been enqueued.  Returns [(lo, hi, ready_op)] with the ranges tiling [0, n_active) from the arena END downwards
    (the backward pass finishes the last layers first), ready_op non-decreasing, so bucket k can be all-reduced while the backward
    of the earlier layers still runs."""
    if not writes:
        return []
    ws = sorted(writes, key=lambda w: -w[0])
    total = sum(hi - lo for lo, hi, _ in ws)
    # the LAST bucket (final only when the backward ends: nothing left to hide its exchange behind) is kept small -- the arena's leading
    # tensors up to 1 % of the parameters -- and the others share the rest equally
    tail = 0
    if n_buckets > 2:
        for lo, hi, _ in reversed(ws):
            if tail + (hi - lo) > total // 100:
                break
            tail += hi - lo
    nbig = n_buckets - (1 if tail > 0 else 0)
    target = (total - tail) / float(max(1, nbig))
    out, acc, ready, hi_edge, left = [], 0, -1, n_active, total
    for i, (lo, hi, r) in enumerate(ws):
        acc += hi - lo
        left -= hi - lo
        ready = max(ready, r)
        last = i == len(ws) - 1
        cut_tail = tail > 0 and left == tail
        if last or cut_tail or (acc >= target and len(out) < nbig - 1):
            edge = 0 if last else lo
            out.append([edge, hi_edge, ready])
            hi_edge, acc = edge, 0
    for k in range(1, len(out)):                 # a later bucket is never launched before an earlier one
        out[k][2] = max(out[k][2], out[k - 1][2])
    return [tuple(b) for b in out]


class NetHandle:
    """awr_net: the checkpoint layout of one backbone.  `layout` = [(key, shape, kind, offset, unused)] in state_dict order."""

    def __init__(self, kind, nstack, J, downsample):
        self.h = C.c_void_p()
        L.call("awr_net_create", kind, nstack, J, downsample, C.byref(self.h))
        nt, npar, nact, nbuf = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        ncnt, nst = C.c_int(), C.c_int()
        L.call("awr_net_sizes", self.h, C.byref(nt), C.byref(npar), C.byref(nact), C.byref(nbuf), C.byref(ncnt), C.byref(nst))
        self.n_params, self.n_active, self.n_buffers, self.n_counters, self.nstage = npar.value, nact.value, nbuf.value, ncnt.value, nst.value
        self.layout = []
        key, kind_, ndim, off, unused = C.c_char_p(), C.c_int(), C.c_int(), C.c_int64(), C.c_int()
        shape = (C.c_int64 * 4)()
        for i in range(nt.value):
            L.call("awr_net_tensor_info", self.h, i, C.byref(key), C.byref(kind_), C.byref(ndim), shape, C.byref(off), C.byref(unused))
            self.layout.append((key.value.decode(), tuple(shape[:ndim.value]), _KIND_NAMES[kind_.value], off.value, bool(unused.value)))

    def bind(self, params, grads, buffers):
        L.call("awr_net_bind", self.h, L.ptr(params), L.ptr(grads), L.ptr(buffers))

    def __del__(self):
        try:
            if self.h:
                L.lib.awr_net_destroy(self.h)
                self.h = None
        except Exception:
            pass


class DpComm:
    """awr_dp: an RCCL communicator owned by the library (include/awr_hip.h "Data-parallel API"; librccl.so is dlopen'ed).  One per
    rank process, on the current device.  `unique_id()` on rank 0 -> ship the 128 bytes to the other ranks -> `DpComm(rank, world, id)`."""

    @staticmethod
    def available():
        v, p = C.c_int(), C.c_char_p()
        ok = L.lib.awr_dp_available(C.byref(v), C.byref(p)) == 0
        return ok, v.value, (p.value or b"").decode()

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        L.call("awr_dp_unique_id", buf)
        return buf.raw

    def __init__(self, rank, world, uid):
        self.h = C.c_void_p()
        self.rank, self.world = rank, world
        L.call("awr_dp_init", int(rank), int(world), C.create_string_buffer(bytes(uid), 128), C.byref(self.h))

    @classmethod
    def from_process_group(cls, pg):
        """Bootstrap over an existing torch.distributed group (any backend): rank 0's id is broadcast as an object."""
        rank, world = torch.distributed.get_rank(pg), torch.distributed.get_world_size(pg)
        box = [cls.unique_id() if rank == 0 else None]
        torch.distributed.broadcast_object_list(box, src=torch.distributed.get_global_rank(pg, 0) if hasattr(torch.distributed, "get_global_rank") else 0, group=pg)
        return cls(rank, world, box[0])

    def allreduce(self, t):
        L.call("awr_dp_allreduce", self.h, L.ptr(t), t.numel(), L.stream())

    def broadcast(self, t, root=0):
        L.call("awr_dp_broadcast", self.h, L.ptr(t), t.numel(), int(root), L.stream())

    def wait(self):
        L.call("awr_dp_wait", self.h, L.stream())

    def close(self):
        if self.h:
            L.lib.awr_dp_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Plan:
    """awr_plan + its boundary tensors.  Building allocates every buffer; replaying allocates nothing and never synchronises."""

    def __init__(self, net, B, H, F, J, training, supervised, bn_repeat=1, n_buckets=1):
        dev = net.device
        self.net, self.B, self.H, self.dev, self.training = net, B, H, dev, bool(training)
        self.bn_repeat, self.n_buckets = bn_repeat, n_buckets
        nstage = net.nstage
        self.img = torch.zeros(B, 1, H, H, device=dev)
        self.outputs = [torch.zeros(B, 4 * J, F, F, device=dev) for _ in range(nstage)]
        self.grad_outs = [torch.zeros(B, 4 * J, F, F, device=dev) if training else None for _ in range(nstage)]
        self.gen = 0
        mask = 0
        for s in (range(nstage) if supervised == "all" else supervised):
            mask |= 1 << s
        outs = (C.c_void_p * nstage)(*[o.data_ptr() for o in self.outputs])
        gouts = (C.c_void_p * nstage)(*[g.data_ptr() for g in self.grad_outs]) if training else None
        self.h = C.c_void_p()
        L.call("awr_plan_create", net._handle.h, B, H, int(bool(training)), mask, bn_repeat, n_buckets, L.ptr(self.img), outs, gouts, C.byref(self.h))
        nbytes, det, nf, nb, nbk, ng, nbn = C.c_int64(), C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        L.call("awr_plan_info", self.h, C.byref(nbytes), C.byref(det), C.byref(nf), C.byref(nb), C.byref(nbk), C.byref(ng), C.byref(nbn))
        self.bytes = nbytes.value + sum(t.numel() * 4 for t in [self.img] + self.outputs + [g for g in self.grad_outs if g is not None])
        self.det, self.n_ops, self.n_gemm, self.n_bn = bool(det.value), (nf.value, nb.value), ng.value, nbn.value
        self.buckets = []
        lo, hi, ready = C.c_int64(), C.c_int64(), C.c_int()
        for i in range(nbk.value):
            L.call("awr_plan_bucket", self.h, i, C.byref(lo), C.byref(hi), C.byref(ready))
            self.buckets.append((lo.value, hi.value, ready.value))
        self.macs, self._ops, self._side_ok = {}, ([], []), set()
        name, macs, flags = C.c_char_p(), C.c_double(), C.c_int()
        for lst in (0, 1):
            for i in range(self.n_ops[lst]):
                L.call("awr_plan_op", self.h, lst, i, C.byref(name), C.byref(macs), C.byref(flags))
                nm = name.value.decode()
                self._ops[lst].append((nm, bool(flags.value & 2)))
                if flags.value & 2:
                    self.macs[nm] = macs.value
                if flags.value & 1:
                    self._side_ok.add(nm)
        self.tuned = {}
        self.n_side, self._hook, self._cb = 0, None, None
        self.nhwc, self._hook_exc = False, None

    # ---- introspection ---------------------------------------------------------------------------------------
    def op_names(self, which):
        return [n for n, _ in self._ops[0 if which in (0, "fwd", "forward") else 1]]

    def tensors(self, lazy=False):
        """Parity / debugging introspection (awr_plan_tensor): {name: (value, grad or None)} torch views (NHWC, no copy) of the plan's
        activation buffers in build order.  "<layer>.out" is a conv's raw output (before its BatchNorm).  lazy=True: also the
        BatchNorm(+ReLU) outputs that are never written -- {name: (value buffer of the tensor they normalise, None, scale[C], shift[C],
        relu)}: the tensor they stand for is [relu](value * scale + shift).  Gradients are what the last backward replay left behind."""
        out, i = {}, 0
        name, lz = C.c_char_p(), C.c_int()
        dims = (C.c_int * 4)()
        buf, grad, sc, sh = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()

        def view(p, shape):
            n = 1
            for d in shape:
                n *= d
            holder = type("DevMem", (), {"__cuda_array_interface__": {"shape": (n,), "typestr": "<f4", "data": (p, False), "version": 2}})()
            return torch.as_tensor(holder, device=self.dev).view(*shape)
        while L.lib.awr_plan_tensor(self.h, i, C.byref(name), dims, C.byref(buf), C.byref(grad), C.byref(lz), C.byref(sc), C.byref(sh)) == 0:
            i += 1
            if not buf.value:
                continue
            shape = tuple(dims)
            if lz.value:
                if lazy:
                    out[name.value.decode()] = (view(buf.value, shape), None, view(sc.value, shape[3:]), view(sh.value, shape[3:]), lz.value == 2)
                continue
            out[name.value.decode()] = (view(buf.value, shape), view(grad.value, shape) if grad.value else None)
        return out

    def set_dp(self, dp):
        """Attach (or detach: None) a library-owned RCCL communicator: the backward replay all-reduces every gradient bucket itself
        (no Python in the loop, no ctypes trampoline)."""
        L.call("awr_plan_set_dp", self.h, dp.h if dp is not None else None)
        self._dp = dp

    # ---- NHWC boundary ----------------------------------------------------------------------------------------
    def head_nhwc(self, stage):
        """(pred pointer, gradient pointer or None, Cp): stage's dense map as the head GEMM leaves it, (B, F*F, Cp) rows."""
        pred, grad, cp = C.c_void_p(), C.c_void_p(), C.c_int()
        L.call("awr_plan_head_nhwc", self.h, int(stage), C.byref(pred), C.byref(grad), C.byref(cp))
        return pred.value, grad.value, cp.value

    def set_nhwc_boundary(self, on):
        """on: the plan stops transposing to / from the NCHW boundary tensors (`outputs` / `grad_outs` are then neither written nor
        read); the caller runs the NHWC head / loss kernels on head_nhwc()'s buffers.  Returns False when the plan cannot serve it
        (a supervised stage whose gradient has a second producer inside the network)."""
        if L.lib.awr_plan_set_nhwc_boundary(self.h, int(bool(on))) != 0:
            return False
        self.nhwc = bool(on)
        return True

    def dense_map(self, stage):
        """The reference-layout (B, 4J, F, F) dense map of `stage` (transposed on demand while the NHWC boundary is on)."""
        out = self.outputs[stage]
        if self.nhwc:
            pred, _, cp = self.head_nhwc(stage)
            L.call("awr_nhwc_to_nchw", pred, self.B, out.shape[2] * out.shape[3], cp, out.shape[1], L.ptr(out), L.stream())
        return out

    # ---- streams / buckets -----------------------------------------------------------------------------------
    def set_streams(self, n_side, comm=False):
        """n_side library-owned HIP streams: weight-gradient GEMMs of the backward are issued round-robin on them and run beside
        the data-gradient chain; forked forward branches (ResNet downsample projections, Hourglass skip residuals) use them too.
        comm: one more stream that gradient buckets are handed to (data parallel)."""
        L.call("awr_plan_set_streams", self.h, int(n_side), int(bool(comm)))
        self.n_side = int(n_side)

    @property
    def bucket_hook(self):
        return self._hook

    @bucket_hook.setter
    def bucket_hook(self, fn):
        """fn(lo, hi): gradient arena [lo, hi) is final in the order of the CURRENT torch stream (the engine switches torch to the
        stream the native runner names, so `torch.distributed.all_reduce(..., async_op=True)` orders itself correctly)."""
        self._hook = fn
        if fn is None:
            self._cb = None
            L.call("awr_plan_set_bucket_callback", self.h, None, None)
            return

        def trampoline(user, lo, hi, stream):
            # ctypes swallows exceptions raised inside a callback (it prints "Exception ignored" and the native replay carries on): a
            # failed collective would leave this bucket un-reduced and the replicas would silently diverge.  Stash the first error;
            # run_backward() re-raises it once the native call has returned.
            if self._hook_exc is not None:
                return
            try:
                if (stream or 0) == L.stream():          # the stream the plan is being replayed on: torch is already there
                    fn(lo, hi)
                else:                                     # the plan's comm stream
                    with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=self.dev)):
                        fn(lo, hi)
            except BaseException as e:                    # noqa: BLE001 -- anything: it must not be lost
                self._hook_exc = e
        self._cb = L.BUCKET_CB(trampoline)          # keep the ctypes thunk alive as long as the plan may call it
        L.call("awr_plan_set_bucket_callback", self.h, C.cast(self._cb, C.c_void_p), None)

    # ---- replay ----------------------------------------------------------------------------------------------
    def refresh_weights(self):
        L.call("awr_plan_refresh_weights", self.h, L.stream())

    def run_forward(self):
        L.call("awr_plan_forward", self.h, L.stream())

    def run_backward(self):
        self._hook_exc = None
        L.call("awr_plan_backward", self.h, L.stream())
        if self._hook_exc is not None:
            e, self._hook_exc = self._hook_exc, None
            raise L.AwrError("gradient-bucket hook failed during the backward replay (the bucket was NOT reduced): %r" % (e,)) from e

    def forward(self):
        self.run_forward()
        if self.training:
            self.net._counters += self.bn_repeat

    def backward(self):
        self.run_backward()

    def timed(self, which, every=False):
        """Serial replay of one launch list with a HIP-event pair around every launch -> {name: seconds} of the conv / stem launches
        (`every`: of all launches, BatchNorm / pooling / element-wise ones included)."""
        lst = 0 if which in (0, "fwd", "forward") else 1
        ms = (C.c_float * self.n_ops[lst])()
        L.call("awr_plan_run_timed", self.h, lst, L.stream(), ms)
        out, self.timed_counts = {}, getattr(self, "timed_counts", {})
        for i, (n, g) in enumerate(self._ops[lst]):
            if g or (every and ms[i] > 0):      # (GEMM-family names are unique; the element-wise launches share theirs: summed, counted)
                if n not in out:
                    self.timed_counts[n] = 0
                out[n] = out.get(n, 0.0) + ms[i] * 1e-3
                self.timed_counts[n] += 1
        return out

    # ---- autotune --------------------------------------------------------------------------------------------
    def _gemm(self, i):
        name, tm, tn, tb, us, tuned = C.c_char_p(), C.c_int(), C.c_int(), C.c_int(), C.c_float(), C.c_int()
        L.call("awr_plan_gemm", self.h, i, C.byref(name), C.byref(tm), C.byref(tn), C.byref(tb), C.byref(us), C.byref(tuned))
        algo = C.c_int()
        L.call("awr_plan_gemm_algo", self.h, i, C.byref(algo))
        return name.value.decode(), (tm.value, tn.value, tb.value, algo.value), us.value, bool(tuned.value)

    def autotune(self, reps=3, cache_key=None):
        """Pick the fastest workgroup tile (and split-K depth) for every GEMM launch of this static plan by timing the candidates
        in place (natively, HIP events).  Runs right after a first replay (buffers hold real data; the next step rebuilds whatever
        the tuner scribbles on).  Shapes never change, so this is a one-off cost of a few hundred milliseconds per plan; with
        AWR_TUNE_CACHE=<file> the choices are stored / reloaded so that a later process (e.g. a profiler run of the same command)
        skips the timing.  No-op in deterministic mode."""
        if self.det:
            return
        cache_file = os.environ.get("AWR_TUNE_CACHE")
        if cache_key:
            # tile choices differ between the modes (products, staging, and the plan's accumulation order: blocked doubles the accumulators)
            cache_key += "/x%d/s%d/a%d" % (L.lib.awr_get_gemm_products(), L.lib.awr_get_gemm_staging(), int(getattr(self, "accum", 0)))
        names = [self._gemm(i)[0] for i in range(self.n_gemm)]
        if cache_file and cache_key and os.path.exists(cache_file):
            try:
                ent = json.load(open(cache_file)).get(cache_key)
            except (OSError, ValueError):
                ent = None
            if ent and all(n in ent for n in names):
                for i, n in enumerate(names):
                    (tm, tn, tb, *rest), t = ent[n]
                    algo = rest[0] if rest else 0     # (files written before the algorithm was stored: the plan's own choice stays)
                    if tm and tn:                     # (0, 0): a launch the tuner leaves alone (one-geometry kernels)
                        L.call("awr_plan_set_gemm", self.h, i, tm, tn, tb, float(t))
                        if algo:
                            L.call("awr_plan_set_gemm_algo", self.h, i, algo)
                    self.tuned[n] = ((tm, tn, tb, algo), t)
                return
        L.call("awr_plan_autotune", self.h, int(reps), L.stream())
        for i in range(self.n_gemm):
            n, tile, us, _ = self._gemm(i)
            self.tuned[n] = (tile, round(us, 1))
        if cache_file and cache_key:
            try:
                allc = json.load(open(cache_file)) if os.path.exists(cache_file) else {}
            except (OSError, ValueError):
                allc = {}
            allc[cache_key] = {k: [list(v[0]), v[1]] for k, v in self.tuned.items()}
            try:
                json.dump(allc, open(cache_file, "w"))
            except OSError:
                pass

    def __del__(self):
        try:
            if self.h and self.net._handle.h:
                L.lib.awr_plan_destroy(self.h)
            self.h = None
        except Exception:
            pass
