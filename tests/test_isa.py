"""CPU tier (needs hipcc, which cross-compiles without a GPU): ISA-level checks of the cross-workgroup hand-off.

The hand-off publishes a queue entry after write-through stores of the env's state; the consumer usually runs on another XCD,
whose L2 is not coherent with the producer's.  The stores only count once they are *acknowledged*: there must be an
`s_waitcnt vmcnt(0)` between the last state store and the atomic that reserves / publishes the queue slot
(mujoco_jaco_amd/csrc/include/jaco/wave_ops.h: dev_stores_done)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SRC = r'''
#include <hip/hip_runtime.h>
#include <jaco/wave_ops.h>
// the publication sequence of queue_push (physics_kernel.h), on its own
__global__ void handoff(float* state, int* remaining, int* count, int* list, int env, int left) {
  const int lane = lane_id();
  st_wt(&state[env * 64 + lane], (float)lane);                 // the env's state rows: write-through
  if (lane == 0) st_wt_i(&remaining[env], left);
  dev_stores_done();
  wave_sync();
  if (lane == 0) {
    int slot = jaco_atomic_inc(count);
    st_wt_i(&list[slot], env);
    dev_stores_done();
  }
}
'''


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_handoff_waits_for_store_acknowledgement(tmp_path):
    src = tmp_path / "handoff.hip"
    src.write_text(SRC)
    out = tmp_path / "handoff.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                           "-I", os.path.join(ROOT, "mujoco_jaco_amd", "csrc", "include"), "-o", str(out), str(src)])
    isa = [l.strip() for l in out.read_text().splitlines() if re.match(r"\s+[a-z_]+", l) and not l.strip().startswith((".", ";"))]
    stores = [i for i, l in enumerate(isa) if l.startswith("global_store") and "sc1" in l]
    atomics = [i for i, l in enumerate(isa) if l.startswith("global_atomic_add")]
    assert len(stores) >= 3 and len(atomics) == 1, (stores, atomics)
    state_stores = [i for i in stores if i < atomics[0]]
    assert state_stores, "the state stores must precede the slot reservation"
    waits = [i for i, l in enumerate(isa) if l.startswith("s_waitcnt") and "vmcnt(0)" in l]
    assert any(state_stores[-1] < w < atomics[0] for w in waits), "no s_waitcnt vmcnt(0) between the state stores and the queue atomic"
    # ... and the entry store itself is acknowledged before the workgroup goes on (light_left is decremented after it)
    entry_store = [i for i in stores if i > atomics[0]]
    assert entry_store and any(w > entry_store[-1] for w in waits)
