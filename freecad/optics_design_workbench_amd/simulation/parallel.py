"""Multi-GPU: ray-index sharding + one sum-reduction of the detector results.

The reference's only parallelism is N independent `FreeCAD -c` worker
processes tracing their own random rays and merging result files
(simulation/processes/simulation_loop.py:386-396, 450-507).  Here rays are
addressed by a global Philox counter, so a job of n rays is split into
disjoint index ranges, one per GPU (one process per GPU, torch.distributed);
tracing needs no communication, and the per-detector histogram and the
counters are summed once at the end (RCCL reduce over xGMI; `gloo` on CPU in
the tests).  Integer sums make the result independent of the GPU count.
"""

WARM_BASE = 1 << 44   # warm-up steps use ray indices far away from the job's


def shardRange(first, n, rank, world):
  """[first, first+n) split into `world` contiguous ranges; -> (first_r, n_r)"""
  base, rem = divmod(int(n), int(world))
  n_r = base + (1 if rank < rem else 0)
  first_r = int(first) + rank * base + min(rank, rem)
  return first_r, n_r


def shardFirst(step, rank, world, n_per, warm=False):
  """weak scaling: step s of rank r traces [((s*world)+r)*n_per, +n_per)"""
  return (WARM_BASE if warm else 0) + (step * world + rank) * n_per


def reduceResults(tracer, dist, torch, dst=0):
  """sum counters + histogram of every rank into rank `dst`'s device buffers with ONE reduce: the library keeps
  both in one block of int64 words (`Tracer.resultsView`, odw_device_results), of which this is a zero-copy view"""
  tracer.sync()
  view, _ = tracer.resultsView()
  if hasattr(view, '__cuda_array_interface__'):
    t = torch.as_tensor(view, device=torch.device('cuda', tracer.device))
    if dist.get_backend() == 'nccl':
      dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM)
    else:
      # a group without a device backend (gloo: ranks that share one GPU, a rehearsal of a node on one): the block
      # travels through host memory, still as one reduce
      h = t.cpu()
      dist.reduce(h, dst=dst, op=dist.ReduceOp.SUM)
      if dist.get_rank() == dst:
        t.copy_(h)
    torch.cuda.synchronize()
  else:                                  # a block in host memory (the CPU stand-in of the tests, gloo)
    dist.reduce(torch.from_numpy(view), dst=dst, op=dist.ReduceOp.SUM)


class Ranks:
  """the process group of a multi-GPU `runSimulation`: one process per GPU
  (torch.distributed; backend nccl = RCCL on the GPU box, gloo in the CPU
  tests).  The only exchanges are a handful of int64 totals per launch (so
  that every rank evaluates the end criteria on the job's totals) and the
  run-folder name; hit rows never travel between ranks -- each rank writes its
  own `*-hits.pkl` files into the shared run folder, which is how the
  reference's worker processes merge their results
  (simulation_loop.py:450-507, freecad_document.py:1491-1504)."""

  def __init__(self, dist=None, device=None):
    self.dist = dist
    self.rank = dist.get_rank() if dist is not None else 0
    self.world = dist.get_world_size() if dist is not None else 1
    self._device = device
    # RCCL runs a collective on the device of its tensors, and a rank's communicator belongs to one
    # device.  The tracer's device is that device: settled HERE, before any GPU work and on every rank
    # alike -- a check inside the first collective would fail on some ranks only (rank 0's device 0 is
    # always "current") and leave the others waiting in all_reduce until the watchdog ends the job.
    if dist is not None and self.world > 1 and device is not None and dist.get_backend() == 'nccl':
      import torch
      if torch.cuda.current_device() != int(device):
        torch.cuda.set_device(int(device))

  @classmethod
  def detect(cls, dist=None, device=None):
    """`dist` given: use it; otherwise an initialised torch.distributed group is
    picked up when the process was started by a launcher (WORLD_SIZE > 1)"""
    import os
    import sys
    if dist is None and int(os.environ.get('WORLD_SIZE', '1')) > 1 and 'torch' in sys.modules:
      import torch.distributed as td
      if td.is_available() and td.is_initialized():
        dist = td
    return cls(dist, device)

  def _tensor_device(self):
    import torch
    if self.dist.get_backend() == 'nccl':
      # the tracer's device (made current in __init__), else the process' current device
      return torch.device('cuda', int(self._device) if self._device is not None else torch.cuda.current_device())
    return torch.device('cpu')

  def sum(self, values):
    """element-wise sum of a short list of integers over all ranks"""
    if self.world == 1:
      return [int(v) for v in values]
    import torch
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=self._tensor_device())
    self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
    return [int(v) for v in t.cpu()]

  def sumFloats(self, values):
    """element-wise sum of a float64 vector over all ranks, on every rank (the sweep's
    value -> result table: each rank contributes its own entries, zeros elsewhere)"""
    import numpy as np
    values = np.asarray(values, dtype=np.float64)
    if self.world == 1:
      return values.copy()
    import torch
    t = torch.from_numpy(values.copy()).to(self._tensor_device())
    self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
    return t.cpu().numpy()

  def broadcast(self, obj, src=0):
    if self.world == 1:
      return obj
    box = [obj if self.rank == src else None]
    self.dist.broadcast_object_list(box, src=src)
    return box[0]

  def shard(self, first, n):
    return shardRange(first, n, self.rank, self.world)

  def barrier(self):
    if self.world > 1:
      self.dist.barrier()
