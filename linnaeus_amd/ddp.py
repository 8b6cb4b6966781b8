"""Data-parallel training of the HIP mFormerV1: one process per GPU, RCCL over xGMI.

Replaces what the reference gets from torch DistributedDataParallel (linnaeus/main.py:936-983):
parameters are broadcast from rank 0 once, and every step the gradients are averaged across
ranks.  Because the model's backward is four native segments writing into ONE flat fp32
gradient arena ordered by segment, each segment is exactly one contiguous all-reduce bucket:

    backward segment 0 (tail + RoPE stage 4 + downsample 3)   ~52 % of the gradient bytes
    backward segment 1 (RoPE stage 3 + downsample 2)          ~42 %
    backward segment 2 (ConvNeXt stage 2 + downsample 1)      ~ 5 %
    backward segment 3 (ConvNeXt stage 1 + stem)              ~ 1 %
    (the metadata heads' backward runs on the plan's side stream and is joined one segment later: the stage-4 heads
    travel with segment 1's bucket, the stage-3 heads with segment 2's -- joining them in their own segment exposed
    ~1.5 ms of small fp32 GEMMs per step)

As soon as a segment's kernels are enqueued, its bucket's all-reduce is issued asynchronously:
the process group runs it on ITS OWN internal HIP stream behind an event it records on the
compute stream (torch ProcessGroupNCCL; backend "nccl" is RCCL on ROCm), so the collective of
segment k runs under the compute of segments k+1.. and only the last ~1 % of the bytes is
exposed; `finish()` makes the compute stream wait for the returned work handles.  No stream of
our own sits between the two (until round 4 one did: a fifth stream on a device with four
hardware queues -- launch, metadata heads, weight gradients, ours, the process group's; now
launch + weight gradients (which also carries the metadata heads) + the process group's = 3).
xGMI is point-to-point; the bucket count is deliberately tiny (4 large messages) so RCCL can
pipeline each over all 7 links.

`GradBucketReducer` holds the collective logic and works on any device (the gloo/CPU tests
drive it directly); `DataParallel` wires it to a model.
"""
from __future__ import annotations

import contextlib
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist


class GradBucketReducer:
    """Average contiguous slices ("buckets") of one flat gradient tensor across ranks."""

    def __init__(self, arena: torch.Tensor, bounds: Dict[int, Tuple[int, int]], process_group=None, compress_bf16: bool = False,
                 single_rank_collectives: bool = False, collective=None, telemetry: bool = False):
        self.arena = arena
        # measurement seam: `collective(buf)` replaces dist.all_reduce and runs on a communication stream of this object (tools/bench_cu_hog.py
        # puts a stand-in kernel there that occupies CUs and HBM the way a ring all-reduce would, on one GPU)
        self.collective = collective
        self.force = single_rank_collectives  # issue the collectives even with one rank (exercises the stream logic on one GPU)
        self.bounds = dict(bounds)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.compress = compress_bf16
        self.cuda = arena.is_cuda
        # only the stand-in collective needs a stream of ours; a real one runs on the process group's own stream
        self.comm_stream = torch.cuda.Stream(device=arena.device) if (self.cuda and collective is not None) else None
        self._pending = []   # (work handle, what to do on the compute stream once it is done)
        self._scratch = {}
        backend = dist.get_backend(process_group) if dist.is_initialized() else ""
        self._avg = backend == "nccl"  # ReduceOp.AVG exists for NCCL/RCCL only
        # telemetry (bench.py's N > 1 line, a few untimed steps): per bucket an event pair issue -> done; `done` is recorded on a probe
        # stream that waits for the work handle, so it costs a stream -- off in every timed step
        self.telemetry = bool(telemetry)
        self._probe = None
        self._events = {}

    def streams_used(self) -> int:
        """HIP streams this object itself owns (the process group's internal stream not counted)."""
        return int(self.comm_stream is not None) + int(self._probe is not None)

    def reduce_bucket(self, key: int) -> None:
        """Issue the all-reduce of one bucket.  On GPU it starts behind everything the current (compute) stream has enqueued so far and
        runs beside what is enqueued afterwards; nothing here blocks the host."""
        if (self.world == 1 and not self.force) or key not in self.bounds:
            return
        lo, hi = self.bounds[key]
        if hi <= lo:
            return
        buf = self.arena[lo:hi]
        if not self.cuda:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg)
            buf.div_(self.world)
            return
        if self.collective is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                self.collective(buf)
            return
        ev0 = None
        if self.telemetry:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record(torch.cuda.current_stream())
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        if self.compress:
            sc = self._scratch.get(key)
            if sc is None:
                sc = self._scratch[key] = torch.empty(hi - lo, dtype=torch.bfloat16, device=buf.device)
            sc.copy_(buf)  # on the compute stream; the collective is ordered behind it
            work = dist.all_reduce(sc, op=op, group=self.pg, async_op=True)

            def post(buf=buf, sc=sc):
                buf.copy_(sc)
                if not self._avg:
                    buf.div_(self.world)
        else:
            work = dist.all_reduce(buf, op=op, group=self.pg, async_op=True)

            def post(buf=buf):
                if not self._avg:
                    buf.div_(self.world)
        self._pending.append((work, post))
        if ev0 is not None:
            if self._probe is None:
                self._probe = torch.cuda.Stream(device=buf.device)
            ev1 = torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(self._probe):
                work.wait()  # (stream-level: the probe stream waits for the process group's stream)
                ev1.record(self._probe)
            self._events.setdefault(key, []).append((ev0, ev1))

    def finish(self) -> None:
        """Make the compute stream wait for every issued collective (no host sync)."""
        if not self.cuda:
            return
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        for work, post in self._pending:
            work.wait()  # ProcessGroupNCCL: the CURRENT stream waits for the collective's end event; the host goes on
            post()
        self._pending.clear()

    def bucket_report(self):
        """[{bucket, bytes, issue_to_done_ms (mean over the recorded steps), steps}] from the telemetry events; synchronises the device."""
        if not self._events:
            return []
        torch.cuda.synchronize()
        out = []
        for key in sorted(self._events):
            lo, hi = self.bounds[key]
            ms = [a.elapsed_time(b) for a, b in self._events[key]]
            out.append({"bucket": key, "bytes": (hi - lo) * (2 if self.compress else 4), "issue_to_done_ms": round(sum(ms) / len(ms), 3), "steps": len(ms)})
        self._events.clear()
        return out


def broadcast_module_state(module: torch.nn.Module, src: int = 0, process_group=None) -> None:
    """Rank `src`'s parameters and buffers to everyone (DDP's construction-time broadcast)."""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    with torch.no_grad():
        seen = set()
        for t in list(module.parameters()) + list(module.buffers()):
            if id(t) in seen:
                continue
            seen.add(id(t))
            dist.broadcast(t.data, src=src, group=process_group)


class DataParallel(torch.nn.Module):
    """Wrap a linnaeus_amd mFormerV1 for one-process-per-GPU data parallelism."""

    def __init__(self, module: torch.nn.Module, process_group=None, compress_bf16: bool = False, broadcast: bool = True,
                 single_rank_collectives: bool = False, collective=None, cu_margin: int = 0, meta_stream: int = 2):
        super().__init__()
        self.module = module
        self.collective = collective
        self.telemetry = False
        # stream budget (four hardware queues): the metadata-head chains move from their own side stream onto the weight-gradient stream,
        # so that a step holds launch + weight-gradient streams + the process group's collective stream (meta_stream=1: keep the side stream)
        if hasattr(module, "set_meta_stream"):
            module.set_meta_stream(meta_stream)
        if cu_margin:
            # persistent kernels launch on (CUs - margin) workgroups: spares the dispatcher a queue of workgroups that cannot be placed
            # beside the collective's.  Not needed for throughput -- they draw their tiles from atomic counters, so a late workgroup owes
            # nothing (profiles/r04_cu_hog.log) -- hence 0 by default.
            from . import _lib as L

            L.check(L.lib().lnx_set_cu_margin(int(cu_margin)), "lnx_set_cu_margin")
        self.force = single_rank_collectives  # test hook: run the collectives with one rank too
        self.pg = process_group
        self.compress = compress_bf16
        self._reducer: Optional[GradBucketReducer] = None
        self._sync = True
        module.grad_mode = "direct"
        module._segment_hook = self._on_segment
        if broadcast:
            broadcast_module_state(module, 0, process_group)

    def _on_segment(self, seg: int) -> None:
        if not self._sync:
            return
        m = self.module
        if self._reducer is None or self._reducer.arena.data_ptr() != m._grad_arena.data_ptr():
            self._reducer = GradBucketReducer(m._grad_arena, m._segment_bounds, self.pg, self.compress, single_rank_collectives=self.force, collective=self.collective)
        self._reducer.telemetry = self.telemetry
        self._reducer.reduce_bucket(seg)
        if seg == 3:
            self._reducer.finish()

    def stream_budget(self) -> dict:
        """HIP streams a data-parallel step of this model issues work on: the device has four hardware queues by default
        (GPU_MAX_HW_QUEUES), and a fifth busy stream serialises forked work behind unrelated kernels."""
        import os

        m = self.module
        meta = getattr(m, "_meta_stream", None)
        wg = bool(getattr(m, "_wgrad_stream", True)) and os.environ.get("LNX_WGRAD_STREAM") != "0"
        names = ["launch"]
        if wg or meta == 2:
            names.append("weight-gradient" + (" (+ metadata heads)" if meta == 2 else ""))
        if meta in (None, 1) and os.environ.get("LNX_NO_SIDE_STREAM") is None:
            names.append("metadata heads")
        own = self._reducer.streams_used() if self._reducer is not None else int(self.collective is not None)
        names += ["reducer (stand-in collective / telemetry probe)"] * own
        names.append("process group's collective stream (ProcessGroupNCCL internal)")
        return {"streams": names, "count": len(names), "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "unset (driver default 4)")}

    def bucket_report(self):
        return self._reducer.bucket_report() if self._reducer is not None else []

    @contextlib.contextmanager
    def no_sync(self):
        """Skip the gradient all-reduce (gradient-accumulation micro-steps, train.py:172-196)."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)
