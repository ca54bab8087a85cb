"""Multi-GPU layout: independent environment shards, one observation gather per rollout step.

Environments never interact (no cross-env term anywhere in env_script/env_mujoco*.py), so rank r of W owns the
contiguous env range [r*B/W, (r+1)*B/W) and steps it with no data-path collective.  The only exchange is the
concatenation of the per-rank observation rows ([B_local, 26] f32 = 6.8 MB at 65 536 envs) once per step:
torch.distributed.all_gather_into_tensor, i.e. RCCL over xGMI with backend "nccl" (gloo on CPU in the tests).
"""
import torch
import torch.distributed as dist


def shard_range(rank, world, total):
    """Contiguous env range [lo, hi) of `rank`; the remainder goes to the first ranks."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_seed(base_seed, rank):
    """Distinct counter-based RNG streams per rank (seeds rank * 2^32 + i)."""
    return (int(base_seed) + (int(rank) << 32)) & 0xFFFFFFFFFFFFFFFF


class ObsGather:
    """Pre-allocated all_gather of equally sized observation shards.

    `gather(local)` is the plain blocking form.  `start(local)` / `wait()` is the overlapped form for a rollout loop on a GPU: the
    local rows are copied into a staging buffer on the caller's stream (the env step's own stream, so the copy is ordered behind
    the kernels that wrote them and the rows may be overwritten by the next step right away), and the collective itself is issued
    on a side stream -- RCCL's transfers over xGMI then run under the next step's launch set instead of in front of it.  `wait()`
    makes the caller's stream wait for the last started gather and returns the concatenated rows [world * local_rows, width]."""

    def __init__(self, local_rows, width, device, dtype=torch.float32, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.out = torch.empty(self.world * local_rows, width, device=device, dtype=dtype)
        self.device = torch.device(device)
        self._overlap = self.device.type == "cuda" and self.world > 1
        self.timing, self._timed = False, []   # enable_timing(): event pairs around each collective on the side stream
        if self._overlap:
            self.stage = torch.empty(local_rows, width, device=device, dtype=dtype)
            self.side = torch.cuda.Stream(device=device)
            self._ready = torch.cuda.Event()
            self._done = None

    def __call__(self, local_obs):
        if self.world == 1:
            self.out.copy_(local_obs)
        else:
            dist.all_gather_into_tensor(self.out, local_obs.contiguous(), group=self.group)
        return self.out

    def start(self, local_obs):
        if not self._overlap:
            self(local_obs)
            return
        cur = torch.cuda.current_stream(self.device)
        if self._done is not None:
            cur.wait_event(self._done)          # the previous gather has read the staging buffer (it finished a whole step ago)
        self.stage.copy_(local_obs)
        self._ready.record(cur)
        try:
            with torch.cuda.stream(self.side):
                self.side.wait_event(self._ready)
                t0 = None
                if self.timing:
                    t0 = torch.cuda.Event(enable_timing=True); t0.record(self.side)
                dist.all_gather_into_tensor(self.out, self.stage, group=self.group)
                self._done = torch.cuda.Event(enable_timing=self.timing)
                self._done.record(self.side)
                if t0 is not None:
                    self._timed.append((t0, self._done))
        except RuntimeError as e:   # a backend that cannot issue the collective from a side stream: same collective, caller's stream, said out loud
            import sys
            print("ObsGather: overlapped gather unavailable (%s); using the blocking form" % (str(e).splitlines()[0],), file=sys.stderr)
            self._overlap, self._done = False, None
            self(local_obs)

    def enable_timing(self, on=True):
        """Time every collective started from here on with an event pair on the side stream (what the gather itself costs, as
        opposed to what it adds to a step: it runs under the next step's launch set)."""
        self.timing, self._timed = bool(on), []

    def gather_time_ms(self):
        """(mean, max, count) of the timed collectives' device time in ms; synchronises the side stream.  None if nothing was timed."""
        if not self._timed:
            return None
        self.side.synchronize()
        ms = [a.elapsed_time(b) for a, b in self._timed]
        return sum(ms) / len(ms), max(ms), len(ms)

    def wait(self):
        if self._overlap and self._done is not None:
            torch.cuda.current_stream(self.device).wait_event(self._done)
        return self.out
