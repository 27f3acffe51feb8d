import json, sys
l=json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
print({k:l[k] for k in ("value","ms_per_step")}); r=l["roofline"]; print({k:r.get(k) for k in ("kernel","avg_launch_ms","achieved","frac","frac_of_attainable","attainable_read_GBps","traffic","traffic_source","frac_of_min")}); print(r.get("second_kernel"))
print("hbm_physical", (l.get("hbm_physical") or {}).get("frac_of_peak"), "unbracketed", (l.get("without_event_bracketing") or {}).get("value"), "lowsync", (l.get("mgs2_lowsync") or {}).get("value"))
sw=l.get("sharded_world1") or {}
print("sharded", {k:sw.get(k) for k in ("value","ms_per_step","xsync","error")}, (sw.get("collectives") or {}).get("per_iteration"))
for k,v in (l.get("configs") or {}).items(): print(k, v["value"], v.get("value_min"), v.get("value_max"), v.get("ms_per_block_step"), (v.get("roofline") or {}).get("kernel"), (v.get("roofline") or {}).get("frac"), {kk_:vv.get("frac") for kk_,vv in list(v.get("kernels",{}).items())[:3]})
print((l.get("parity") or {}).get("ok"), (l.get("parity") or {}).get("ritz_relerr"), (l.get("cpu_baseline") or {}).get("value"), (l.get("cpu_baseline") or {}).get("cores"))
