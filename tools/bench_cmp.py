import json,sys
for f in sys.argv[1:]:
    j=json.loads(open(f).read().strip().split("\n")[-1]); print(f, j["value"], "match solo", j["stage_ms_one_chunk_in_flight"]["match"], "total solo", j["stage_ms_one_chunk_in_flight"]["total"])
