import sys, json
tag = sys.argv[1]
for ln in sys.stdin:
    d = json.loads(ln); p = d["profiled_seconds"]
    print(tag, "rank", d["rank"], "wall", round(d["wall_ms_no_comm"], 1), "L", round(d["L_GB_this_rank"], 1), "arena", round(d["arena_GB_this_rank"], 1),
          "upd3", round(p["update_wave_tiles"], 3), round(p["update_wave_tiles_TF"], 1), "ea", round(p["extend_add+zero"], 3))
