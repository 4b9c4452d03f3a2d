import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("headline", round(d["value"],1), "p50", round(d["latency_ms"]["p50"],3), {k:round(v*1e3,3) for k,v in d["latency_ms"]["stage_seconds_last_query"].items()})
for k in ("image_in","image_in_1080p"):
    if k in d and d[k]:
        v=d[k]; print(k, round(v["value"],1), "p50", round(v["latency_ms"]["p50"],3), v["identical_to_single_flight"], v["oracle_end_to_end"]["frames_identical"], {a:round(b,3) for a,b in v["path_stage_ms_one_frame_alone"].items()}, "extract", round(v["stage_ms_one_frame_alone"]["extract(K9, incl. the count's synchronisation)"],3), v["frames_localised"])
