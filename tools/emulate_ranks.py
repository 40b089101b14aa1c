#!/usr/bin/env python3
"""One rank at a time of an N-rank split of the headline frame, through the whole N > 1 step of bench.py on ONE GPU
(`bench.py --emulate-rank r/N`): ms per step, kernel time per launch, host enqueue time per step, against the 1-GPU step.
    python tools/emulate_ranks.py [band_rows]      (all 8 ranks of N = 8, ranks 0 and 1 of N = 4 and N = 2)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
br = sys.argv[1] if len(sys.argv) > 1 else "8"
def run(extra):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup", "60", "--no-cpu-baseline"] + extra,
                       capture_output=True, text=True)
    return json.loads(p.stdout.strip().splitlines()[-1])
one = run(["--timed-only"])
print("1 GPU: %.4f ms/step" % one["ms_per_step"], flush=True)
for n, ranks in ((8, range(8)), (4, (0, 1)), (2, (0, 1))):
    worst = 0.0
    for r in ranks:
        d = run(["--emulate-rank", "%d/%d" % (r, n), "--band-rows", br])
        worst = max(worst, d["ms_per_step"])
        print("N=%d rank %d (bands of %s rows, %d rows): %.4f ms/step, kernel %.4f ms per launch (two frames in flight), host enqueue %.4f ms/step, "
              "oracle sample of its bands: %s" % (n, r, br, d["emulated_rank"]["rows"], d["ms_per_step"], d["kernel_ms_per_launch"],
                                                  d["host_enqueue_ms_per_step"], d.get("gathered_frame_matches_oracle_on_sample")), flush=True)
    print("N=%d: slowest emulated rank %.4f ms/step -> %.1f %% of the 1-GPU step / %d" % (n, worst, 100.0 * one["ms_per_step"] / (n * worst), n), flush=True)
