"""Static instruction mix of one kernel by source function (diagnostic).

Input: the device ISA of the HIP library compiled with line tables
   hipcc --offload-arch=gfx950 -O3 ... --cuda-device-only -S -g1 jaco_env.hip -o build/jaco_env_g1.s
Every instruction is charged to the source function whose body holds its innermost .loc line (inlined code keeps its own lines).
Counts are static (each instruction once, whatever its loop trip count); straight-line stages of the substep loop read directly.
   usage: isa_by_function.py build/jaco_env_g1.s [kernel-symbol-substring]
"""
import collections, re, sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CS = os.path.join(ROOT, "mujoco_jaco_amd", "csrc")


def function_ranges(path):
    """[(first line, name)] of the function definitions of a source file (JDEV / template / __global__ heads at column 0)."""
    out = []
    pat = re.compile(r"^(?:template\s*<[^>]*>\s*)?(?:JDEV|static|__global__|inline|extern)\b.*?\b([A-Za-z_][A-Za-z_0-9]*)\s*\(")
    for i, l in enumerate(open(path), 1):
        mm = pat.match(l)
        if mm and not l.rstrip().endswith(";"):
            out.append((i, mm.group(1)))
    return out


HELPERS = set("mk3 ld3 st3 ld4 dot cross norm normalized ldm stm mul mulT col quat2mat ldsv stsv cross_motion cross_force inert_mul comp_add comp_advance "
              "ld_frame st_frame m_index joint_sincos axis_rot fmul_rn f16_round make_frame".split())


def main():
    asm = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "_Z19jaco_physics_kernel12JacoStepArgs"
    files, ranges = {}, {}
    kinds = collections.OrderedDict([("valu", 0), ("salu", 0), ("lds", 0), ("vmem", 0), ("scratch", 0), ("smem", 0), ("mfma", 0), ("wait", 0), ("branch", 0), ("other", 0)])
    per = collections.defaultdict(lambda: collections.Counter())
    inside, cur = False, ("?", 0)
    last_stage = "?"
    for l in open(asm, errors="replace"):
        t = l.strip()
        if t.startswith(".file"):
            mm = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', t)
            if mm: files[int(mm.group(1))] = os.path.basename(mm.group(3) or mm.group(2))
            continue
        if not t or t.startswith(";"): continue
        lab = re.match(r"^([A-Za-z_$][\w$.]*):", t)
        if lab:
            inside = want in lab.group(1)
            continue
        if t.startswith(".loc"):
            p = t.split()
            cur = (files.get(int(p[1]), "?"), int(p[2]))
            continue
        if not inside or t.startswith("."): continue
        op = t.split()[0]
        if op.startswith("v_mfma") or op.startswith("v_smfmac"): k = "mfma"
        elif op.startswith("v_"): k = "valu"
        elif op.startswith("ds_"): k = "lds"
        elif op.startswith("scratch_"): k = "scratch"
        elif op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"): k = "vmem"
        elif op.startswith("s_load") or op.startswith("s_buffer_load"): k = "smem"
        elif op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"): k = "wait"
        elif op.startswith("s_cbranch") or op.startswith("s_branch"): k = "branch"
        elif op.startswith("s_"): k = "salu"
        else: k = "other"
        f, ln = cur
        if f not in ranges:
            pth = os.path.join(CS, f) if os.path.exists(os.path.join(CS, f)) else os.path.join(CS, "include", "jaco", f)
            ranges[f] = function_ranges(pth) if os.path.exists(pth) else []
        name = "?"
        for first, nm in ranges[f]:
            if first <= ln: name = nm
            else: break
        key = f + ":" + name
        # small helpers (vector algebra, wave ops, libm wrappers) are inlined everywhere: charge them to the enclosing stage, taken
        # to be the last non-helper function seen (instruction scheduling blurs stage borders a little)
        helper = f in ("__clang_hip_math.h", "wave_ops.h", "amd_warp_functions.h", "amd_hip_atomic.h", "amd_device_functions.h") or name in HELPERS or name == "?"
        if "--leaf" not in sys.argv:
            if helper: key = last_stage
            else: last_stage = key
        per[key][k] += 1
    tot = collections.Counter()
    for c in per.values(): tot.update(c)
    print("%-44s" % "function" + "".join("%8s" % k for k in kinds))
    for fn, c in sorted(per.items(), key=lambda kv: -kv[1]["valu"]):
        print("%-44s" % fn[:44] + "".join("%8d" % c[k] for k in kinds))
    print("%-44s" % "TOTAL" + "".join("%8d" % tot[k] for k in kinds))


if __name__ == "__main__":
    main()
