"""Static instruction mix of the light step kernel by call site inside run_env (diagnostic).

The ISA listing (hipcc ... --cuda-device-only -S -g1) carries the inline stack of every .loc as a comment
   ; ./physics_kernel.h:1019:8 @[ ./physics_kernel.h:1629:20 @[ ./physics_kernel.h:1912:12 @[ ... ] ] ]
Each instruction is charged to the frame of that stack that lies inside run_env (the stage call it belongs to), so inlined helpers count
for the stage that called them.  Counts are static.   usage: isa_by_callsite.py build/jaco_env_xxx.s [kernel symbol]
"""
import collections, re, sys, os
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SRC = os.path.join(ROOT, "mujoco_jaco_amd", "csrc", "physics_kernel.h")


def main():
    asm = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "_Z19jaco_physics_kernel12JacoStepArgs"
    src = open(SRC).read().split("\n")
    lo = next(i for i, l in enumerate(src, 1) if l.startswith("JDEV int run_env("))
    hi = next(i for i, l in enumerate(src, 1) if i > lo and l.startswith("// light tier: one workgroup"))
    per = collections.defaultdict(collections.Counter)
    inside, site = False, 0
    for l in open(asm, errors="replace"):
        t = l.strip()
        lab = re.match(r"^([A-Za-z_$][\w$.]*):", t)
        if lab:
            inside = lab.group(1) == want
            continue
        if not inside or not t: continue
        if t.startswith(".loc"):
            frames = re.findall(r"physics_kernel\.h:(\d+):", t)
            inner = [int(x) for x in frames if lo <= int(x) < hi]
            site = inner[-1] if inner else (-1 if frames else site)   # outermost frame inside run_env
            continue
        if t[0] in ".;": continue
        op = t.split()[0]
        k = ("mfma" if op.startswith("v_mfma") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "scratch" if op.startswith("scratch_")
             else "vmem" if op.startswith(("global_", "flat_", "buffer_")) else "smem" if op.startswith("s_load") else "nop" if op.startswith("s_nop")
             else "wait" if op.startswith("s_waitcnt") else "salu" if op.startswith("s_") else "other")
        per[site][k] += 1
        if op in ("v_readlane_b32", "v_writelane_b32"): per[site]["lane-xfer"] += 1
    kinds = ["valu", "lane-xfer", "salu", "lds", "vmem", "scratch", "smem", "mfma", "nop", "wait"]
    print("%6s %-70s" % ("line", "run_env source") + "".join("%9s" % k for k in kinds))
    tot = collections.Counter()
    for site, c in sorted(per.items(), key=lambda kv: -kv[1]["valu"]):
        tot.update(c)
        if c["valu"] < 15: continue
        text = src[site - 1].strip()[:68] if site > 0 else "(outside run_env)"
        print("%6d %-70s" % (site, text) + "".join("%9d" % c[k] for k in kinds))
    print("%6s %-70s" % ("", "TOTAL") + "".join("%9d" % tot[k] for k in kinds))


if __name__ == "__main__":
    main()
