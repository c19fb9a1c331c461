import re, sys
lines = open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/parity_report.txt').read().splitlines()
nok = 0
for l in lines:
    m = re.search(r': ([0-9.e+-]+|nan|inf) \(([<>]) ([0-9.e+-]+)\)$', l)
    if m:
        v, op, b = float(m.group(1)), m.group(2), float(m.group(3))
        ok = v < b if op == '<' else v > b
        if not ok:
            print("FAIL", l)
        else:
            nok += 1
    else:
        print("    ", l)
print("passing checks:", nok)
