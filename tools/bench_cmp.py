import json,sys
for f in sys.argv[1:]:
    j=json.loads(open(f).read().strip().split("\n")[-1]); print(f, j["value"], "match solo", (j.get("stage_ms_one_chunk_in_flight") or j["stage_ms"])["match"], "total solo", (j.get("stage_ms_one_chunk_in_flight") or j["stage_ms"])["total"])
