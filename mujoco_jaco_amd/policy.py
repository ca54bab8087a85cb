"""Reader + forward pass of the reference's shipped HPC policies.

The reference evaluates `models_baseline/policies/<task>/policy.zip` through `HPC.predict_subgoal(obs, deterministic=True)`
(main.py:221-263).  HPC lives in the TMmichi fork of stable-baselines, which is absent from /root/reference and from this
image [EXT]; what IS in the tree is the file format (a stable-baselines zip: `data` JSON, `parameter_list`, `parameters`
= an .npz of TensorFlow variables) and, inside `data`, the pickled `primitives` table: per tail its observation index,
action index, `obs_relativity`, `act_scale` and layer names (main.py:97-168 builds the same table).  This module restates
the forward pass from that table:

  * primitive tail  (`level1_reaching/level0`, `level1_grasping/level0`, ...): SAC MlpPolicy actor [EXT stable-baselines
    sac/policies.py]: x = obs[obs_index] (or obs[ref] - obs[tar] under `obs_relativity: subtract`), 3 x (dense + relu),
    mu = dense, log_std = clip(dense_1, -20, 2).
  * weight tail     (`level1_<task>/weight`): same trunk ([256, 256, 256]) on its own observation slice, softmax over one
    logit per primitive (`dense`); `dense_1` (one unit) is not used by the deterministic action.
  * composition [EXT, HPC paper / MCP]: the composite action distribution is the weighted product of the primitives'
    Gaussians, per action dimension over the primitives that own that dimension:
        1/sigma^2 = sum_i w_i / sigma_i^2,     mu = sigma^2 * sum_i w_i mu_i / sigma_i^2,
    and the deterministic action is tanh(mu) (action space [-1, 1]^7, so no rescaling).

Nothing here touches the oracle.  The weights travel as data: either the reference-format zip itself or the .npz that
tests/golden/make_policy_vectors.py extracts from it.
"""
import base64
import io
import json
import pickle
import zipfile

import numpy as np
import torch

LOG_STD_MIN, LOG_STD_MAX = -20.0, 2.0


class _Box:
    """Stand-in for gym.spaces.Box while unpickling the primitives table (gym is not a dependency)."""

    def __setstate__(self, d):
        self.__dict__.update(d)


class _Opaque:
    """Stand-in for objects the table carries but the forward pass never reads (the Box's numpy RandomState)."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, d):
        self.state = d


def _allowed_globals():
    import collections
    import copyreg
    import _codecs
    try:
        from numpy._core import multiarray as ma   # numpy >= 2
    except ImportError:   # pragma: no cover
        from numpy.core import multiarray as ma
    table = {}
    for mod in ("numpy.core.multiarray", "numpy._core.multiarray"):
        table[(mod, "_reconstruct")] = ma._reconstruct
        table[(mod, "scalar")] = ma.scalar
    table[("numpy", "ndarray")] = np.ndarray
    table[("numpy", "dtype")] = np.dtype
    for name in ("__randomstate_ctor", "__bit_generator_ctor", "__generator_ctor"):
        table[("numpy.random._pickle", name)] = _Opaque
    for name in ("dict", "list", "tuple", "set", "frozenset", "int", "float", "str", "bytes", "bool", "slice", "range", "complex"):
        table[("builtins", name)] = getattr(__import__("builtins"), name)
    table[("collections", "OrderedDict")] = collections.OrderedDict
    table[("copyreg", "_reconstructor")] = copyreg._reconstructor
    table[("_codecs", "encode")] = _codecs.encode
    return table


class _Unpickler(pickle.Unpickler):
    """Exact (module, name) allowlist: the `primitives` blob of a policy.zip is untrusted input (a pickle), and anything beyond the
    handful of constructors the reference's own zips use -- numpy array / dtype reconstruction, gym's Box, plain containers -- is
    refused.  No module prefix is trusted: `builtins.eval`, `numpy.testing...` etc. raise."""
    _table = None

    def find_class(self, module, name):
        if module in ("gym.spaces.box", "gym.spaces", "gym.spaces.space") and name in ("Box", "Space"):
            return _Box
        if _Unpickler._table is None:
            _Unpickler._table = _allowed_globals()
        try:
            return _Unpickler._table[(module, name)]
        except KeyError:
            raise pickle.UnpicklingError("policy.zip primitives table: refusing global %s.%s" % (module, name))


def _tail_from_params(P, prefix):
    """Collect fc*/dense/dense_1 of one TF variable scope `prefix` (e.g. 'model/pi/level1_reaching/level0')."""
    hidden, i = [], 0
    while "%s/fc%d/kernel:0" % (prefix, i) in P:
        hidden.append((P["%s/fc%d/kernel:0" % (prefix, i)], P["%s/fc%d/bias:0" % (prefix, i)]))
        i += 1
    return {"hidden": hidden, "out": (P[prefix + "/dense/kernel:0"], P[prefix + "/dense/bias:0"]),
            "out2": (P[prefix + "/dense_1/kernel:0"], P[prefix + "/dense_1/bias:0"])}


def read_policy_zip(path):
    """The reference's policy.zip -> plain dict of numpy arrays + index lists (no TensorFlow, no gym)."""
    z = zipfile.ZipFile(path)
    data = json.loads(z.read("data"))
    P = dict(np.load(io.BytesIO(z.read("parameters"))))
    if not data.get("tails"):
        # a plain (non-composite) SAC MlpPolicy zip, e.g. models_baseline/policies/reaching/policy.zip: one actor under model/pi.
        # Its observation slice is not in the zip in a loadable form (policy_kwargs pickles a TensorFlow op); the reference wires
        # this very policy as the `reaching` primitive with obs_relativity subtract ref [17..22] - tar [1..6] and actions [0..5]
        # (main.py:97-103), which is also how it was trained on task `reaching` (goal pose minus EE pose, 6 inputs).
        t = _tail_from_params(P, "model/pi")
        nin, nout = t["hidden"][0][0].shape[0], t["out"][0].shape[1]
        if nin != 6 or nout != 6:
            raise ValueError("plain policy zip with %d inputs / %d actions: only the reaching layout (6 / 6) is known" % (nin, nout))
        t.update(name="model/pi", obs_index=[1, 2, 3, 4, 5, 6, 17, 18, 19, 20, 21, 22], act_index=[0, 1, 2, 3, 4, 5],
                 rel_ref=[17, 18, 19, 20, 21, 22], rel_tar=[1, 2, 3, 4, 5, 6], act_scale=1.0, is_weight=False)
        return {"task": "reaching", "tails": [t]}
    prim = _Unpickler(io.BytesIO(base64.b64decode(data["primitives"][":serialized:"]))).load()
    tails = []
    for name in data["tails"]:
        item = prim[name]
        t = _tail_from_params(P, "model/pi/" + name)
        rel = (item.get("obs_relativity") or {}).get("subtract")
        t.update(name=name, obs_index=list(item["obs"][1]), act_index=list(item["act"][1]),
                 rel_ref=list(rel["ref"]) if rel else [], rel_tar=list(rel["tar"]) if rel else [],
                 act_scale=float(item["act_scale"]) if item.get("act_scale") is not None else 1.0,
                 is_weight=name.split("/")[-1] == "weight")
        tails.append(t)
    return {"task": data.get("composite_primitive_name"), "tails": tails}


def save_npz(policy, path):
    out = {"task": np.array(policy["task"] or ""), "ntails": np.array(len(policy["tails"]))}
    for i, t in enumerate(policy["tails"]):
        p = "t%d_" % i
        out[p + "name"] = np.array(t["name"]); out[p + "is_weight"] = np.array(int(t["is_weight"])); out[p + "act_scale"] = np.array(t["act_scale"])
        for k in ("obs_index", "act_index", "rel_ref", "rel_tar"):
            out[p + k] = np.array(t[k], np.int64)
        out[p + "nhidden"] = np.array(len(t["hidden"]))
        for j, (W, b) in enumerate(t["hidden"]):
            out[p + "W%d" % j] = W; out[p + "b%d" % j] = b
        out[p + "oW"], out[p + "ob"] = t["out"]; out[p + "o2W"], out[p + "o2b"] = t["out2"]
    np.savez_compressed(path, **out)


def load_npz(path):
    Z = np.load(path)
    tails = []
    for i in range(int(Z["ntails"])):
        p = "t%d_" % i
        tails.append({"name": str(Z[p + "name"]), "is_weight": bool(int(Z[p + "is_weight"])), "act_scale": float(Z[p + "act_scale"]),
                      "obs_index": Z[p + "obs_index"].tolist(), "act_index": Z[p + "act_index"].tolist(),
                      "rel_ref": Z[p + "rel_ref"].tolist(), "rel_tar": Z[p + "rel_tar"].tolist(),
                      "hidden": [(Z[p + "W%d" % j], Z[p + "b%d" % j]) for j in range(int(Z[p + "nhidden"]))],
                      "out": (Z[p + "oW"], Z[p + "ob"]), "out2": (Z[p + "o2W"], Z[p + "o2b"])})
    return {"task": str(Z["task"]), "tails": tails}


class HPCPolicy:
    """Batched deterministic forward of a composite HPC policy on a torch device.

    relativity_sign: +1 feeds obs[ref] - obs[tar] to a tail with `obs_relativity: subtract`, -1 the opposite (the fork's code
    is absent; +1 = "goal minus current pose" is the convention that makes the reaching primitive approach its goal).
    """

    def __init__(self, policy, device="cpu", relativity_sign=1.0, nact=7):
        self.device = torch.device(device)
        self.task = policy["task"]
        self.nact = nact
        self.relativity_sign = float(relativity_sign)
        t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=self.device)
        idx = lambda a: torch.tensor(a, dtype=torch.long, device=self.device)
        self.tails = []
        for T in policy["tails"]:
            self.tails.append({"name": T["name"], "is_weight": T["is_weight"], "act_scale": T["act_scale"],
                               "obs_index": idx(T["obs_index"]), "act_index": idx(T["act_index"]),
                               "rel_ref": idx(T["rel_ref"]), "rel_tar": idx(T["rel_tar"]), "has_rel": len(T["rel_ref"]) > 0,
                               "hidden": [(t(W), t(b)) for W, b in T["hidden"]], "out": (t(T["out"][0]), t(T["out"][1])),
                               "out2": (t(T["out2"][0]), t(T["out2"][1]))})
        self.primitives = [T for T in self.tails if not T["is_weight"]]
        self.weight_tail = next((T for T in self.tails if T["is_weight"]), None)

    @classmethod
    def load(cls, path, **kw):
        return cls(load_npz(path) if str(path).endswith(".npz") else read_policy_zip(path), **kw)

    @staticmethod
    def _trunk(T, x):
        for W, b in T["hidden"]:
            x = torch.relu(x @ W + b)
        return x

    def tail_input(self, T, obs):
        if T["has_rel"]:
            return self.relativity_sign * (obs[:, T["rel_ref"]] - obs[:, T["rel_tar"]])
        return obs[:, T["obs_index"]]

    @torch.no_grad()
    def predict(self, obs):
        """obs [B, 26] -> (action [B, nact] in [-1, 1], weight [B, nprimitives])."""
        obs = torch.as_tensor(obs, dtype=torch.float32, device=self.device)
        if obs.dim() == 1:
            obs = obs[None]
        B = obs.shape[0]
        if self.weight_tail is not None:
            h = self._trunk(self.weight_tail, self.tail_input(self.weight_tail, obs))
            logw = torch.log_softmax(h @ self.weight_tail["out"][0] + self.weight_tail["out"][1], dim=1)
        else:
            logw = torch.zeros(B, len(self.primitives), device=self.device)
        # precision weights w_i / sigma_i^2 in log form, normalised per action dimension by the largest owner: a weight that
        # underflows in fp32 (the placing policy runs at w = [6e-14, 1]) must not turn a dimension only that primitive owns
        # (the gripper) into 0 / 0 -- there the weight cancels and the composite mean is the primitive's own
        NEG = -1.0e30
        logp = torch.full((len(self.primitives), B, self.nact), NEG, device=self.device)
        mus = torch.zeros(len(self.primitives), B, self.nact, device=self.device)
        for i, T in enumerate(self.primitives):
            h = self._trunk(T, self.tail_input(T, obs))
            mus[i][:, T["act_index"]] = (h @ T["out"][0] + T["out"][1]) * T["act_scale"]
            log_std = torch.clamp(h @ T["out2"][0] + T["out2"][1], LOG_STD_MIN, LOG_STD_MAX)
            logp[i][:, T["act_index"]] = logw[:, i:i + 1] - 2.0 * log_std
        p = torch.exp(logp - logp.max(0, keepdim=True).values)
        mu = (p * mus).sum(0) / p.sum(0)
        w = torch.exp(logw)
        return torch.tanh(mu), w

    @torch.no_grad()
    def predict_graphed(self, obs):
        """predict() replayed from a captured HIP graph (torch.cuda.CUDAGraph = hipGraph on ROCm): the forward pass is ~70 launches of a
        few microseconds each, launch-bound at any batch size.  (Measured on MI355X / ROCm 7.2: the replay takes as long as the eager launches,
        ~1.5 ms of a 53 ms policy-driven step at 65 536 envs -- bench.py keeps the eager call.)  Captured on first use for the batch size seen; a later call with another
        batch size re-captures.  obs must be a [B, 26] tensor on this policy's (GPU) device; the returned tensors are the graph's static
        outputs, overwritten by the next call."""
        obs = torch.as_tensor(obs, dtype=torch.float32, device=self.device)
        g = getattr(self, "_graph", None)
        if g is None or g[1].shape != obs.shape:
            static_in = obs.clone()
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):          # warm-up off the capture (allocator, lazy module init)
                for _ in range(2):
                    self.predict(static_in)
            torch.cuda.current_stream(self.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.predict(static_in)
            g = self._graph = (graph, static_in, out)
        g[1].copy_(obs)
        g[0].replay()
        return g[2]
