"""Data parallelism for the three-stream batch: one process per GPU, ``torch.distributed`` over RCCL/xGMI.

The reference is single-device.  The step shards naturally over examples, with ONE subtlety
(SURVEY.md §8e): the feature-matching / contrasting losses apply a non-linear function to the *batch mean* of
the discriminator features, so ranks must exchange feature sums in the forward pass (a few hundred bytes to a
few hundred KB) -- averaging per-rank losses or gradients as a stock DDP wrapper would do is NOT the
reference's arithmetic.  Everything is expressed as SUMs:

* forward: all-reduce(sum) of per-rank feature sums, then divide by the global batch; its backward is the
  identity on the local sum (every rank already holds dL/d(mean));
* per-example means (labeled loss, gradient penalty) are local sums / global batch;
* backward: all-reduce(sum) of each network's flat gradient arena, in large buckets (xGMI is point-to-point,
  7 links x ~153 GB/s per GPU: few large messages, not one per tensor), ASYNCHRONOUSLY: ``GradientExchange`` starts a
  bucket as soon as the last backward pass of the step can no longer touch it (the tape reports the frontier,
  ``tape.backward(..., grad_ready=...)``), so the collectives run under the rest of that backward pass and under
  whatever the step does next; the optimizer update waits for them only when the weights are needed again
  (``Experiment.finish_update``).

The same object drives the CPU oracle over gloo in the world-size-2 tests (it only needs tensors).
"""
import os

import torch
import torch.distributed as dist

BUCKET_ELEMENTS = 32 * 1024 * 1024     # 128 MiB fp32 per all-reduce call at most
MIN_BUCKET_ELEMENTS = 2 * 1024 * 1024  # start a bucket once >= 8 MiB of the arena is final (below that a ring all-reduce
#                                        over xGMI is latency-bound; the remainder goes out with finish())


class HipCodec:
    """What ``GradientExchange`` does to a bucket besides the collective, through the C ABI (device tensors only; the CPU
    world-size-2 tests hand it a codec of their own): ``stage`` fills a staging buffer from a slice of the fp32 gradient
    arena -- as bf16 (``srgan_pack_bf16``, round to nearest even) or as a plain copy -- and zeroes its padding,
    ``unstage`` writes the reduced values back."""

    @staticmethod
    def stage(source, buffer):
        from . import _lib, functional as F
        count = source.numel()
        if buffer.dtype == torch.bfloat16:
            _lib.check(_lib.library().srgan_pack_bf16(source.data_ptr(), buffer.data_ptr(), count, _lib.stream_handle()),
                       'srgan_pack_bf16')
            if buffer.numel() > count:          # (the padding is a whole number of 16-byte groups only when count % 8 == 0)
                buffer[count:].zero_()
        else:
            F._unary_raw(F.U_COPY, source, out=buffer[:count])
            if buffer.numel() > count:
                F.fill_(buffer[count:], 0.0)

    @staticmethod
    def unstage(buffer, target):
        from . import _lib, functional as F
        if buffer.dtype == torch.bfloat16:
            _lib.check(_lib.library().srgan_unpack_bf16(buffer.data_ptr(), target.data_ptr(), target.numel(),
                                                        _lib.stream_handle()), 'srgan_unpack_bf16')
        else:
            F._unary_raw(F.U_COPY, buffer[:target.numel()], out=target)


class GradientExchange:
    """Asynchronous sum of one flat gradient buffer over the ranks, from its END towards its start.

    Parameters sit in the arena in forward order, so the backward pass finishes the tail first: ``ready_from(o)``
    declares every element at or after offset ``o`` final and starts buckets over the part not yet sent;
    ``finish()`` sends the remainder; ``wait()`` makes the current stream (NCCL) / the host (gloo) wait for all of
    them.  Every rank makes the same calls with the same offsets (they run the same graph), so the collectives match.

    ``wire`` = 'f32' (default: the bucket is reduced in place) or 'bf16': the bucket is packed into a bf16 staging buffer,
    reduced there and unpacked into the fp32 master gradient in ``wait()`` -- half the bytes per link for the
    comm-sensitive configurations (the DCGAN pair in fp16, where a ring over xGMI costs as much as the
    compute); the sum itself is then rounded to bf16 once per ring step.
    ``form`` = 'all_reduce' (default) or 'reduce_scatter': the same sum as reduce-scatter + all-gather over a bucket padded
    to a multiple of the world size -- on the point-to-point xGMI mesh each rank then exchanges S/N with every peer over
    all seven links at once instead of passing S around one ring, and the two halves are separate collectives that
    RCCL schedules independently."""

    def __init__(self, dp, flat, bucket_elements=BUCKET_ELEMENTS, min_bucket_elements=MIN_BUCKET_ELEMENTS, wire='f32',
                 form='all_reduce', codec=None):
        if wire not in ('f32', 'bf16') or form not in ('all_reduce', 'reduce_scatter'):
            raise ValueError(f'gradient exchange: wire {wire!r} / form {form!r}')
        self.dp, self.flat = dp, flat
        self.bucket, self.min_bucket = bucket_elements, min_bucket_elements
        self.wire, self.form = wire, form
        self.codec = codec if codec is not None else HipCodec
        self.sent_from = flat.numel()       # [sent_from, numel) is already on its way
        self.works = []
        self.pending = []                   # (staging buffer, first, stop): unpacked / copied back in wait()
        self.launched = []                  # (start, stop) of every bucket, in launch order (tests / diagnostics)
        # one communication stream (RCCL through either transport): collectives run in issue order
        self.in_order = getattr(dp, 'abi', None) is not None or getattr(dp, 'device_backend', None) == 'nccl'

    def _bucket(self, first, stop):
        """Start the sum of flat[first:stop]."""
        count, world, group = stop - first, self.dp.world_size, self.dp.group
        target = self.flat[first:stop]
        padded = (count + 8 * world - 1) // (8 * world) * (8 * world) if self.form == 'reduce_scatter' else count
        staged = self.wire == 'bf16' or padded != count or (first * 4) % 16 != 0 and self.form == 'reduce_scatter'
        if staged:
            buffer = torch.empty(padded, dtype=torch.bfloat16 if self.wire == 'bf16' else torch.float32, device=target.device)
            self.codec.stage(target, buffer)
            self.pending.append((buffer, first, stop))
        else:
            buffer = target
        abi = getattr(self.dp, 'abi', None)         # the C ABI's RCCL entry points instead of torch.distributed (opt-in)
        if self.form == 'all_reduce':
            self.works.append(abi.all_reduce_async(buffer) if abi is not None else
                              dist.all_reduce(buffer, op=dist.ReduceOp.SUM, group=group, async_op=True))
            return
        shard = torch.empty(padded // world, dtype=buffer.dtype, device=buffer.device)
        if abi is not None:                 # one communication stream: in issue order
            self.works.append(abi.reduce_scatter_async(shard, buffer))
            self.works.append(abi.all_gather_async(buffer, shard))
            self.pending.append((shard, None, None))
            return
        scatter = dist.reduce_scatter_tensor(shard, buffer, op=dist.ReduceOp.SUM, group=group, async_op=True)
        if not self.in_order:               # (gloo runs asynchronous work on several threads)
            scatter.wait()
        else:
            self.works.append(scatter)
        self.works.append(dist.all_gather_into_tensor(buffer, shard, group=group, async_op=True))
        self.pending.append((shard, None, None))          # kept alive until wait()

    def _send(self, start, stop):
        while stop > start:
            first = max(start, stop - self.bucket)
            self._bucket(first, stop)
            self.launched.append((first, stop))
            stop = first
        self.sent_from = min(self.sent_from, start)

    def ready_from(self, offset):
        offset = max(0, min(int(offset), self.sent_from))
        if self.wire != 'f32' or self.form != 'all_reduce':
            # staged buckets start on a 16-byte boundary of the arena (4 elements: what the pack / copy kernels check; the
            # arena's length and the bucket size are multiples of 4, so every later split keeps it); rounding the frontier
            # UP only declares fewer elements final
            offset = min((offset + 3) // 4 * 4, self.sent_from)
        if self.sent_from - offset >= self.min_bucket:
            self._send(offset, self.sent_from)

    def finish(self):
        if self.sent_from > 0:
            self._send(0, self.sent_from)
        return self

    def wait(self):
        for work in self.works:
            work.wait()
        self.works = []
        for buffer, first, stop in self.pending:
            if first is None:
                continue
            self.codec.unstage(buffer, self.flat[first:stop])
        self.pending = []


class _AllReduceSum(torch.autograd.Function):
    """all-reduce(sum) with identity backward, for torch-autograd users (the oracle)."""

    @staticmethod
    def forward(ctx, tensor, group):
        out = tensor.clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
        return out

    @staticmethod
    def backward(ctx, grad):
        return grad, None


class _StreamWork:
    """What an asynchronous collective of ``AbiCommunicator`` returns: ``wait()`` makes the CURRENT stream wait for it (the
    contract of ``torch.distributed``'s NCCL work objects, which ``GradientExchange.wait`` relies on)."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class AbiCommunicator:
    """The data-parallel collectives through the C ABI's RCCL entry points (include/srgan_hip.h, "collectives") instead of
    ``torch.distributed``: one RCCL communicator per rank, created from a unique id that rank 0 makes and the existing
    process group hands to the others (the rendezvous a reference-side caller would do over its own channel).  Every
    collective runs on ONE communication stream (in issue order) that first waits for the issuing compute stream; gradient
    buckets return a work object whose ``wait()`` orders the current stream behind them, the tiny forward all-reduce of the
    feature sums is waited for at once.  Device tensors only, fp32 or bf16."""

    def __init__(self, dp):
        import ctypes
        from . import _lib
        self.lib = _lib.library()
        self.check = _lib.check
        if not self.lib.srgan_comm_available():
            raise RuntimeError('the C ABI could not resolve librccl.so.1')
        identifier = ctypes.create_string_buffer(128)
        if dp.rank == 0:
            self.check(self.lib.srgan_comm_unique_id(identifier), 'srgan_comm_unique_id')
        raw = dp.broadcast_object(identifier.raw if dp.rank == 0 else None)
        self.comm = ctypes.c_void_p()
        self.check(self.lib.srgan_comm_init(ctypes.byref(self.comm), dp.world_size, dp.rank, raw), 'srgan_comm_init')
        world = ctypes.c_int32()
        self.check(self.lib.srgan_comm_world_size(self.comm, ctypes.byref(world)), 'srgan_comm_world_size')
        assert world.value == dp.world_size
        self.world_size = dp.world_size
        self.stream = torch.cuda.Stream()
        self.calls = 0                       # collectives issued (tests / diagnostics)

    @staticmethod
    def _wire(tensor):
        if not tensor.is_cuda or not tensor.is_contiguous() or tensor.dtype not in (torch.float32, torch.bfloat16):
            raise ValueError('ABI collectives take contiguous fp32 / bf16 device tensors')
        return 0 if tensor.dtype == torch.float32 else 1

    def all_reduce_sum_(self, tensor):
        """In place (the feature sums: F floats).  Issued on the communication stream like every other collective of this
        communicator -- one communicator must not be driven from two streams at once, and a gradient bucket may be in flight
        there -- and waited for at once: the current stream continues behind it."""
        wire = self._wire(tensor)
        self._on_communication_stream(lambda stream: self.check(self.lib.srgan_all_reduce_sum(
            self.comm, tensor.data_ptr(), tensor.data_ptr(), tensor.numel(), wire, stream), 'srgan_all_reduce_sum'), tensor).wait()
        return tensor

    def _on_communication_stream(self, launch, *tensors):
        self.calls += 1
        self.stream.wait_stream(torch.cuda.current_stream())
        launch(self.stream.cuda_stream)
        if not torch.cuda.is_current_stream_capturing():      # (a capture's pool is not recycled before the graph is done)
            for tensor in tensors:
                tensor.record_stream(self.stream)
        event = torch.cuda.Event()
        event.record(self.stream)
        return _StreamWork(event)

    def all_reduce_async(self, tensor):
        wire = self._wire(tensor)
        return self._on_communication_stream(lambda stream: self.check(self.lib.srgan_all_reduce_sum(
            self.comm, tensor.data_ptr(), tensor.data_ptr(), tensor.numel(), wire, stream), 'srgan_all_reduce_sum'), tensor)

    def reduce_scatter_async(self, shard, tensor):
        wire = self._wire(tensor)
        assert shard.dtype == tensor.dtype and shard.numel() * self.world_size == tensor.numel()
        return self._on_communication_stream(lambda stream: self.check(self.lib.srgan_reduce_scatter_sum(
            self.comm, tensor.data_ptr(), shard.data_ptr(), shard.numel(), wire, stream), 'srgan_reduce_scatter_sum'), shard, tensor)

    def all_gather_async(self, tensor, shard):
        wire = self._wire(tensor)
        assert shard.dtype == tensor.dtype and shard.numel() * self.world_size == tensor.numel()
        return self._on_communication_stream(lambda stream: self.check(self.lib.srgan_all_gather(
            self.comm, shard.data_ptr(), tensor.data_ptr(), shard.numel(), wire, stream), 'srgan_all_gather'), shard, tensor)

    def broadcast_(self, tensor, source=0):
        """Rank ``source``'s ``tensor`` to every rank, in place; the current stream continues behind it."""
        wire = self._wire(tensor)
        self._on_communication_stream(lambda stream: self.check(self.lib.srgan_broadcast(
            self.comm, tensor.data_ptr(), tensor.numel(), wire, source, stream), 'srgan_broadcast'), tensor).wait()
        return tensor

    def close(self):
        if self.comm is not None and self.comm.value:
            torch.cuda.synchronize()
            self.check(self.lib.srgan_comm_destroy(self.comm), 'srgan_comm_destroy')
            self.comm = None


def _backend_of(group, device_type):
    """Name of the backend that serves ``device_type`` tensors in ``group`` ('nccl', 'gloo', ...) or None."""
    pg = group if group is not None else dist.distributed_c10d._get_default_group()
    try:
        return pg._get_backend(torch.device(device_type)).name().lower()
    except Exception:
        return None


class DataParallel:
    """One process per GPU.  Two planes:

    * CONTROL (rendezvous, the trial-directory / stdin objects, barriers, the bench's max-over-ranks scalars): ``torch.distributed``
      on host tensors when the group has a host backend (``from_environment`` initialises ``cpu:gloo,cuda:nccl``), otherwise
      on device tensors through nccl;
    * DATA (the feature sums, the gradient buckets, the initial weight broadcast): RCCL over xGMI -- by default through the C
      ABI's own entry points (``AbiCommunicator``: one communicator, one communication stream this process owns, so the
      exchanges are capturable and the process runs FOUR streams = the runtime's four hardware queues; with torch's NCCL
      stream next to it a fifth stream aliased a chain onto another's queue: 81.5 against 85.1 images/s on the forced world-1
      line, profiles/r06q_*), or with ``SRGAN_ABI_COLLECTIVES=0`` through ``torch.distributed``'s nccl backend."""
    force = False
    abi = None           # an AbiCommunicator when the exchanges go through the C ABI's RCCL entry points

    def __init__(self, group=None, force=None):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        self.host_backend = _backend_of(group, 'cpu')                 # 'gloo' or None
        self.device_backend = _backend_of(group, 'cuda') if torch.cuda.is_available() else None   # 'nccl', 'gloo' or None
        # ``force``: run every exchange even on a world of one (SRGAN_FORCE_DP=1, ``bench.py --force-dp``).  Two ranks
        # cannot share a device under RCCL, one rank can: this is how a one-GPU box pushes the feature-sum all-reduce, the
        # asynchronous gradient buckets, their stream-side ``wait()`` and the broadcasts through RCCL.
        self.force = bool(int(os.environ.get('SRGAN_FORCE_DP', '0'))) if force is None else bool(force)
        wanted = os.environ.get('SRGAN_ABI_COLLECTIVES')
        if wanted == '1' and torch.cuda.is_available():
            self.use_abi_collectives()
        elif wanted is None and self.device_backend == 'nccl' and self.host_backend is not None:
            # the default device transport where RCCL is the device backend and a host channel can carry the unique id
            # without waking torch's own communicator; a box without librccl keeps torch.distributed, and says so
            try:
                self.use_abi_collectives()
            except Exception as error:
                print(f'[srgan_amd] C-ABI collectives unavailable ({error}); the device collectives stay on torch.distributed',
                      flush=True)

    def use_abi_collectives(self):
        """Route the feature-sum all-reduce, the gradient buckets and the weight broadcast through the C ABI's RCCL entry
        points; rendezvous, object broadcasts and barriers stay on the process group."""
        if self.abi is None:
            self.abi = AbiCommunicator(self)
        return self.abi

    @property
    def transport(self):
        if self.abi is not None:
            return 'RCCL through the C ABI (srgan_all_reduce_sum / _reduce_scatter_sum / _all_gather / _broadcast)'
        return f'torch.distributed ({self.device_backend or self.host_backend})'

    @property
    def active(self):
        return self.world_size > 1 or self.force

    @classmethod
    def from_environment(cls, backend=None, force=None):
        """Initialise from RANK / WORLD_SIZE / MASTER_* (torchrun); ``nccl`` (= RCCL) when a GPU is present."""
        if not dist.is_initialized():
            if backend is None:
                backend = 'nccl' if torch.cuda.is_available() else 'gloo'
            if backend == 'nccl' and 'LOCAL_RANK' in os.environ and torch.cuda.device_count() > 1:
                torch.cuda.set_device(int(os.environ['LOCAL_RANK']))
            if backend == 'nccl' and dist.is_gloo_available():
                # host tensors (objects, barriers, scalars) over gloo, device tensors over RCCL: torch's own RCCL communicator
                # and its stream only come into being if a device tensor is ever handed to torch.distributed
                if os.environ.get('MASTER_ADDR', '') in ('127.0.0.1', 'localhost'):
                    os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')      # (the container's hostname may not resolve)
                backend = 'cpu:gloo,cuda:nccl'
            dist.init_process_group(backend=backend)
        return cls(force=force)

    # ---- batch bookkeeping -----------------------------------------------------------------------------
    def global_batch(self, local_batch):
        return local_batch * self.world_size

    def local_batch(self, global_batch):
        if global_batch % self.world_size:
            raise ValueError(f'global batch {global_batch} is not divisible by {self.world_size} ranks')
        return global_batch // self.world_size

    def shard(self, tensor):
        """This rank's contiguous slice of the leading (batch) dimension."""
        local = self.local_batch(tensor.shape[0])
        return tensor[self.rank * local:(self.rank + 1) * local]

    # ---- collectives -----------------------------------------------------------------------------------
    def all_reduce_sum_(self, tensor):
        if self.abi is not None and tensor.is_cuda:
            return self.abi.all_reduce_sum_(tensor)
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)
        return tensor

    def all_reduce_sum_autograd(self, tensor):
        return _AllReduceSum.apply(tensor, self.group)

    def _scalar_device(self):
        return 'cpu' if self.host_backend is not None else torch.device('cuda', torch.cuda.current_device())

    def all_reduce_sum_float(self, value):
        device = self._scalar_device()
        holder = torch.tensor([value], dtype=torch.float64, device=device)
        dist.all_reduce(holder, op=dist.ReduceOp.SUM, group=self.group)
        return float(holder.item())

    def all_reduce_max_float(self, value):
        device = self._scalar_device()
        holder = torch.tensor([value], dtype=torch.float64, device=device)
        dist.all_reduce(holder, op=dist.ReduceOp.MAX, group=self.group)
        return float(holder.item())

    def all_reduce_sum_var(self, var, reduce_backward=False):
        """Tape version: forward all-reduce of a (small) var.  Backward: the identity when every rank goes on to compute
        the SAME function of the sum (each then holds the full gradient w.r.t. it already); with ``reduce_backward`` the
        gradient is all-reduced as well -- for a sum that feeds rank-specific terms of a loss that is itself a sum over
        ranks (the ``normalize_feature_norm`` branch, whose distance runs over every rank's own rows)."""
        from .tape import Var, Node, grad_enabled
        def summed(tensor):
            tensor = tensor.clone()
            if self.abi is not None and tensor.is_cuda:
                return self.abi.all_reduce_sum_(tensor.contiguous())
            dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)
            return tensor
        data = summed(var.data)
        out = Var(data, requires_grad=grad_enabled() and var.requires_grad)
        if out.requires_grad:
            def backward(g, needs):
                if not reduce_backward:
                    return (g,)
                return (Var(summed(g.data)),)
            out.node = Node((var,), backward, 'all_reduce_sum')
        return out

    def all_reduce_gradients(self, arena):
        """Sum the flat gradient arena over ranks in 128 MiB buckets (blocking form)."""
        self.gradient_exchange(arena).finish().wait()

    wire, form = 'f32', 'all_reduce'        # defaults of gradient_exchange(); set per run (settings.gradient_wire_dtype / _form)

    def gradient_exchange(self, arena, wire=None, form=None):
        """A fresh asynchronous exchange of ``arena.grad`` (see GradientExchange)."""
        return GradientExchange(self, arena.grad, wire=wire or self.wire, form=form or self.form)

    def broadcast_object(self, value, source=0):
        """A small picklable host object from rank ``source`` to every rank (trial directory, stdin commands)."""
        holder = [value if self.rank == source else None]
        dist.broadcast_object_list(holder, src=source, group=self.group, device=torch.device(self._scalar_device()))
        return holder[0]

    def broadcast_parameters(self, arena, source=0):
        """Make every rank start from rank ``source``'s weights."""
        if self.abi is not None and arena.data.is_cuda:
            self.abi.broadcast_(arena.data, source)
        else:
            dist.broadcast(arena.data, src=source, group=self.group)

    def barrier(self):
        """Host-side meeting point of the ranks (the caller synchronises its device around it where it times something)."""
        if self.host_backend is not None:
            dist.all_reduce(torch.zeros(1), group=self.group)       # host tensors: no device transport involved
        else:
            dist.barrier(group=self.group)
