"""Condenses `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stderr saved to a file) into one line per kernel:
python tools/resource_table.py /tmp/lpc_cols.res.txt [filter]"""
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for b in re.split(r"remark: Function Name: ", txt)[1:]:
        name = b.split()[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout
        dem = re.sub(r"\(.*", "", dem.strip()).replace("void ", "")[:100]
        if flt and flt not in dem:
            continue

        def g(key):
            return re.search(re.escape(key) + r": (\d+)", b).group(1)

        print("%-100s S%3s V%3s A%3s scratch %4s occ %s" % (dem, g("SGPRs"), g("VGPRs"), g("AGPRs"),
                                                          g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]")))


if __name__ == "__main__":
    main()
