import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print({k:d.get(k) for k in ("value","ms_per_step","host_issue_ms_per_step","launch")})
print(json.dumps(d.get("timing")))
print(json.dumps(d.get("headline_repeat_at_end")))
for k,v in d.get("configs",{}).items(): print(k, v.get("value"), v.get("ms_per_step"), v.get("launch"), v.get("timing",{}).get("windows_ms_per_step"), v.get("error"))
r=d.get("roofline") or {}
print(r.get("frac"), r.get("traffic"), r.get("traffic_detail"), r.get("lfcc_kernel"))
e=d.get("configs",{}).get("ecapa_bf16_b128",{}).get("roofline",{})
print("ecapa", e.get("frac"), e.get("traffic_detail"))
print(d.get("cpu_baseline"))
