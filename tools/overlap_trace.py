#!/usr/bin/env python3
"""Do the team kernel and the wave-per-read kernel of one batch overlap?  Run under
rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/overlap_trace.py run
then   python tools/overlap_trace.py report <dir>   prints start/end of the extend kernels of the last launches."""
import sys, glob, csv, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    from thermite_amd import capi, synth
    t = synth.synth_reference()
    ix = capi.Index(t)
    a = capi.Aligner(ix, capi.CI_OPTS)
    for stream in (100, 102):
        bases, off, _ = synth.simulate_reads(t, 500000, 91, sub_rate=0.01, indel_rate=0.001, stream=stream)
        a.upload(bases, off)
        for _ in range(3):
            a.run(); a.sync()
        print(stream, a.timings(), flush=True)
else:
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "extend_kernel" in r["Kernel_Name"]]
    t0 = None
    for r in rows[-12:]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t0 is None:
            t0 = s
        nm = r["Kernel_Name"].split("extend_kernel")[1].split("(")[0]
        print("%-28s start %10.3f ms  end %10.3f ms  dur %.3f ms  grid %s wg %s" % (nm, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))))
