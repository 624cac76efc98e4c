#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per dispatch of the render kernel.

A command launches more than one render kernel (the kernel a handle's two- / three-wave trial did not keep) and one
dispatch per scene handle that has no work at all (the first launch sends the kernel ahead once with n_units = 0,
rtc_capi.hip launch()): the summary is of the kernel NAME with the most dispatches (among those that contain `kernel`),
and per counter of the dispatches within a factor of two of that counter's median - the empty dispatch and the first,
estimate-scheduled frame fall out; what is left is the steady state the bench times."""
import csv, glob, sys, collections, statistics
def summarize(dirs, kernel="rtc_render_kernel", report=None):
    per_name = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for d in dirs:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if kernel in row["Kernel_Name"]:
                    per_name[row["Kernel_Name"]][row["Counter_Name"]][(f, row["Dispatch_Id"])] += float(row["Counter_Value"])
    if not per_name:
        return {}
    name = max(per_name, key=lambda n: max(len(v) for v in per_name[n].values()))
    out = {}
    for counter, by_dispatch in per_name[name].items():
        vals = list(by_dispatch.values())
        med = statistics.median(vals)
        kept = [v for v in vals if med == 0 or 0.5 * med <= v <= 2.0 * med] or vals
        out[counter] = sum(kept) / len(kept)
    if report is not None:
        report["kernel"] = name
        report["dispatches"] = max(len(v) for v in per_name[name].values())
        report["other_kernels"] = {n: max(len(v) for v in c.values()) for n, c in per_name.items() if n != name}
    return out
if __name__ == "__main__":
    rep = {}
    for k, v in sorted(summarize(sys.argv[1:], report=rep).items()):
        print(f"{k:32s} {v:18.1f}")
    print("#", rep)
