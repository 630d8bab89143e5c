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


def reduceTensors(dist, tensors, dst=0):
  for t in tensors:
    dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM)


def reduceResults(tracer, dist, torch, dst=0):
  """sum histogram + counters of every rank into rank `dst`'s device buffers
  (zero-copy views of the tracer's HBM buffers, int64)"""
  tracer.sync()
  views = [tracer.countersView()]
  if tracer._det is not None:
    views.append(tracer.histogramView())
  tensors = [torch.as_tensor(v, device=torch.device('cuda', tracer.device)) for v in views]
  reduceTensors(dist, tensors, dst=dst)
  torch.cuda.synchronize()
