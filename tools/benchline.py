import json,sys
d=json.loads(sys.stdin.read()); print(sys.argv[1], "serial", round(d["value"]), "pipelined", round(d["config"].get("pipelined_solves_per_s") or 0), "kernel_ms", round(d["roofline"]["kernel_ms"],4))
