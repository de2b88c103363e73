"""Static instruction mix of the kernels in a `hipcc -S --cuda-device-only` listing:
python tools/instmix.py /tmp/cols.s [name-filter]"""
import collections
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    parts = re.split(r"\n(_Z\w+):[^\n]*\n", txt)
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1].split("s_endpgm")[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem).replace("void ", "")
        if flt not in dem:
            continue
        c = collections.Counter()
        for ln in body.splitlines():
            m = re.match(r"\s+((?:v|s|ds|global|buffer|scratch|flat)_\w+)", ln)
            if m:
                c[m.group(1)] += 1
        grp = lambda p: sum(v for k, v in c.items() if k.startswith(p))
        print(f"{dem[:110]}\n   VALU {grp('v_')} (packed {grp('v_pk_')})  SALU {grp('s_')}  LDS {grp('ds_')}  global {grp('global_')} flat {grp('flat_')}"
              f"  scratch {grp('scratch_')}  barriers {c['s_barrier']}")
        print("   " + ", ".join(f"{k} {v}" for k, v in c.most_common(12)))


if __name__ == "__main__":
    main()
