"""Post-process a `rocprofv3 --kernel-trace --stats -- python3 bench.py` output directory (tools/profile_bench.sh): pick the parent
process (the headline run = the trace with the most step launches), copy its kernel stats and summarise the light kernel's launches
in bench.py's phases.  usage: profile_window.py <rocprof dir> <bench log with the JSON line> <out stats csv> <out window txt>"""
import csv, glob, json, shutil, sys
d, log, out_csv, out_txt = sys.argv[1:5]
line = json.loads([l for l in open(log) if l.startswith("{")][-1])
pre, warm, steps = line["config"]["preroll_steps"], line["warmup"], line["steps"]
census = line["config"].get("flag_census_steps", 0) or 0   # untimed env steps after the timed window (flag census)
best = None
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("jaco_physics_kernel(")]
    if len(rows) == pre + warm + steps + census: best = (f, sorted(rows))   # (the children run other step counts)
f, rows = best
stats = f.replace("kernel_trace.csv", "kernel_stats.csv")
shutil.copy(stats, out_csv)
ms = [(e - s) * 1e-6 for s, e in rows]
assert len(ms) == pre + warm + steps + census, (len(ms), pre, warm, steps, census)
allms = ms
ms = ms[:pre + warm + steps]
mean = lambda v: sum(v) / len(v)
with open(out_txt, "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py   (the default command; this is the parent process = the headline run; the\n"
            "# action-scale, policy and drift legs run in child processes with trace files of their own; reset-time launches use the kernel symbol\n"
            "# jaco_physics_kernel_listed, so every launch below is an env step)\n")
    o.write("# jaco_physics_kernel launches of the parent, in order: %d pre-roll steps, %d warm-up steps, %d timed steps, %d flag-census steps\n" % (pre, warm, steps, census))
    o.write("launches %d\n" % len(ms))
    o.write("pre-roll steps 1-%d mean    %.3f ms\n" % (pre, mean(ms[:pre])))
    o.write("warm-up + timed (%d) mean   %.3f ms\n" % (warm + steps, mean(ms[pre:])))
    o.write("timed window (last %d) mean %.3f ms   <- the window bench.py's HIP events bracket\n" % (steps, mean(ms[-steps:])))
    o.write("flag census (%d untimed steps after the window) mean %.3f ms\n" % (census, mean(allms[pre + warm + steps:]) if census else 0.0))
    o.write("all launches (the --stats row AverageNs) %.3f ms\n" % mean(allms))
    o.write("bench.py line of the same run: roofline.kernel_ms %.3f (HIP events around the light kernel, timed window), step_launch_set_ms %.3f, value %.0f env-steps/s\n"
            % (line["roofline"]["kernel_ms"], line["roofline"]["step_launch_set_ms"], line["value"]))
print(open(out_txt).read())
