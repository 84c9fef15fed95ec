import json,sys
d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"]), "serial", round(d["config"]["single_stream_solves_per_s"]), "kernel_ms", round(d["roofline"]["kernel_ms"],4))
