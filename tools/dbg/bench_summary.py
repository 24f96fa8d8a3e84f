import json,sys
j=json.load(open(sys.argv[1]))
print("value", j["value"], "ms/step", j["ms_per_step"], "frac", j["roofline"]["frac"], "traffic", j["roofline"]["traffic"])
e=j.get("extras") or {}
if "config3" in e:
    print("config3 ms", e["config3"]["ms_per_step"], "refill:", json.dumps(e["config3"].get("refill_beside_the_loop")))
    print("reference_loop", json.dumps(e["reference_loop"]))
    print("b1_long", e["b1_long_context"]["ms_per_step"], "cfg5 rtf", e["config5"]["end_to_end_rtf"], "cfg5_b32", e["config5_b32"]["ms_per_step"])
print("mimi", json.dumps((j.get("mimi") or {}).get("roofline")))
print("dom", json.dumps(j["roofline"].get("dominant_kernels")))
print("cpu", json.dumps(j.get("cpu_baseline")))
