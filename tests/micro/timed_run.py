#!/usr/bin/env python3
"""wall time and peak RSS (self + children, the largest single process) of a command: python tests/micro/timed_run.py out.json cmd..."""
import json, resource, subprocess, sys, time
t0 = time.time()
rc = subprocess.call(sys.argv[2:])
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
json.dump(dict(cmd=" ".join(sys.argv[2:]), rc=rc, wall_s=round(time.time() - t0, 1), max_rss_gb_largest_process=round(ru.ru_maxrss / 1e6, 2)), open(sys.argv[1], "w"))
print(open(sys.argv[1]).read())
sys.exit(rc)
