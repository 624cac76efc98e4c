#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per dispatch of a kernel."""
import csv, glob, sys, collections
def summarize(dirs, kernel="rtc_render_kernel"):
    acc = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            per = collections.defaultdict(float)
            for row in csv.DictReader(open(f)):
                if kernel in row["Kernel_Name"]:
                    per[(row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
            for (disp, name), v in per.items():
                acc[name].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}
if __name__ == "__main__":
    for k, v in sorted(summarize(sys.argv[1:]).items()):
        print(f"{k:32s} {v:18.1f}")
