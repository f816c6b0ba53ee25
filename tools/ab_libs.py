"""A/B two builds of the library on one box: alternates bench.py between libyart_amd.so and the file
given (YART_AMD_LIB), N rounds.   python tools/ab_libs.py yart_amd/libyart_amd_base.so [rounds]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
other = os.path.abspath(sys.argv[1])
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for r in range(rounds):
    for name, lib in (("new", None), ("base", other)):
        env = dict(os.environ)
        if lib:
            env["YART_AMD_LIB"] = lib
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-seconds", "0", "--steps", "10"], env=env,
                             capture_output=True, text=True).stdout
        d = json.loads(out.strip().splitlines()[-1])
        print(name, d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], flush=True)
