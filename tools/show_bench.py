"""prints the fields of a bench.py JSON line that matter when iterating: python tools/show_bench.py gpurun_out/x.json"""
import json
import sys
d = json.load(open(sys.argv[1]))
for k in ("value", "ms_per_step", "dtype", "roofline", "tolerance_compliant", "tolerance_compliant_fwd", "forward_only_b256", "dropin_step", "other_workloads", "kernel_ms_per_step", "block_ms_per_step",
          "cross_attention_block", "padded_layout", "step_executed_tflops_per_gpu", "dp_exchange", "dp_fallback", "cpu_baseline"):
    if d.get(k) is not None:
        print(k, json.dumps(d[k]))
