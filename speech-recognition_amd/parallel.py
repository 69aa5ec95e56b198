"""Multi-GPU: utterances shard across ranks, only Baum-Welch statistics meet.

One process per GPU.  Scoring and the front-end need no collective (every rank
keeps all word models, a few KB).  Training sums the packed statistics records
of all models with ONE all-reduce per EM iteration (RCCL over xGMI through
``torch.distributed``, backend "nccl"; "gloo" on CPU for tests); the M-step is
then replicated on every rank from identical sums, so parameters never need a
broadcast.  The per-utterance loop being sharded is do_estep,
my_hmmlearn/hmm.py:419-433.
"""
import numpy as np
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(lengths, world_size):
    """Contiguous utterance ranges with near-equal frame counts.
    Returns int64 bounds[world_size+1] into the utterance axis."""
    lengths = np.asarray(lengths, dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(lengths)])
    total = cum[-1]
    targets = (np.arange(world_size + 1, dtype=np.float64) * total / world_size)
    bounds = np.searchsorted(cum, targets, side="left").astype(np.int64)
    bounds[0], bounds[-1] = 0, len(lengths)
    return np.maximum.accumulate(bounds)


def shard_utterances(lengths, rank, world_size):
    """(first, last) utterance of this rank's block."""
    b = shard_bounds(lengths, world_size)
    return int(b[rank]), int(b[rank + 1])


def allreduce_stats(stats):
    """Sum the packed statistics over ranks, in place.  fp64, one message for
    all models (S=16, D=39, 6 models: 9228 doubles = 74 KB, latency-bound)."""
    _, n = world()
    if n > 1:
        if stats.is_cuda and dist.get_backend() == "gloo":
            # gloo's own handling of device tensors stalls on buffers of this size when two
            # ranks share a GPU (the rehearsal set-up of a one-GPU box): stage through the host
            host = stats.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            stats.copy_(host)
        else:
            dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    return stats


def allreduce_stats_rccl(ctx, nccl_comm, stats):
    """The same sum through the C ABI (srx_allreduce_stats) for hosts that hold a raw
    ncclComm_t instead of a torch process group; nccl_comm is the communicator's
    address.  Enqueued on the current stream, in place."""
    import ctypes as C
    from ._lib import lib, check
    check(lib.srx_allreduce_stats(ctx.handle, C.c_void_p(int(nccl_comm)), C.c_void_p(stats.data_ptr()),
                                  int(stats.numel()), ctx.stream_ptr()), "srx_allreduce_stats")
    return stats


class BaumWelchTrainer:
    """Batched EM for a bank of models over this rank's shard of utterances.

    utt_model[u] names the model utterance u trains (the label loop of
    train_hmm, HMM_final.py:28-35, done for all labels in one launch)."""

    def __init__(self, bank, data, utt_model=None, priors=None, mstep="full"):
        self.bank, self.data, self.priors, self.mode = bank, data, priors, mstep
        ctx = bank.ctx
        if utt_model is None:
            self.utt_model = None
            self.utt_order = None
        else:
            um = np.asarray(utt_model, dtype=np.int32)
            if um.shape != (data.U,):
                raise ValueError("utt_model must have one entry per utterance")
            if um.size and (um.min() < 0 or um.max() >= bank.M):
                raise ValueError("utt_model entry outside [0, n_models)")
            self.utt_model = ctx.tensor(um)
            # float32: no visiting order -- the matrix-core E-step builds its own on the device
            # (by model, then by length in descending buckets, whole batches of 16 per model, so
            # no batch is walked twice and few matrix columns idle behind a long utterance;
            # rebuilt on every call, about 30 us at 131,072 utterances).  The shape-generic
            # float64 kernels, which keep a model's sums in registers while the model does not
            # change, get a stable sort by model, made once here.
            self.utt_order = None if data.X.dtype == torch.float32 else \
                ctx.tensor(np.argsort(um, kind="stable").astype(np.int64))
        self.stats = torch.zeros((bank.M, bank.stats_size), dtype=torch.float64, device=ctx.device)

    def step(self):
        """One EM iteration; returns the per-model log-probabilities (device
        tensor, computed with the parameters before the update)."""
        # (a transitions-only M-step reads no sum over the features: small jobs skip that pass)
        self.bank.estep(self.data, self.utt_model, self.utt_order, out=self.stats,
                        stats="transitions" if self.mode == "transitions_only" else "all")
        allreduce_stats(self.stats)
        logprob = self.stats[:, 1].clone()
        self.bank.mstep(self.stats, self.priors, self.mode)
        return logprob
