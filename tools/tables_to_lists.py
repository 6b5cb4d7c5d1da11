#!/usr/bin/env python3
"""Turns markdown tables whose rows do not fit 120 columns into nested bullet lists, then re-wraps prose (development tool:
`python tools/tables_to_lists.py FILE...`).  Narrow tables are left as they are."""
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from reflow_md import WIDTH, reflow


def convert(text):
    lines, out, i, fence = text.split("\n"), [], 0, False
    while i < len(lines):
        ln = lines[i]
        if ln.lstrip().startswith("```"):
            fence = not fence
        if (not fence and ln.startswith("|") and i + 1 < len(lines) and re.match(r"^\|[\s:|-]+\|\s*$", lines[i + 1])):
            j = i + 2
            while j < len(lines) and lines[j].startswith("|"):
                j += 1
            block = lines[i:j]
            if max(len(b) for b in block) <= WIDTH:
                out += block
            else:
                hdr = [c.strip() for c in re.split(r"(?<!\\)\|", block[0].strip().strip("|"))]
                for row in block[2:]:
                    cells = [c.strip().replace("\\|", "|") for c in re.split(r"(?<!\\)\|", row.strip().strip("|"))]
                    out.append("- **" + cells[0].strip("*") + "**")
                    for h, c in zip(hdr[1:], cells[1:]):
                        if c:
                            out.append("  - " + (h + ": " if h else "") + c)
            i = j
            continue
        out.append(ln)
        i += 1
    return reflow("\n".join(out))


if __name__ == "__main__":
    for path in sys.argv[1:]:
        with open(path) as fh:
            src = fh.read()
        res = convert(src)
        with open(path, "w") as fh:
            fh.write(res)
        print(path, "max line", max(len(l) for l in res.split("\n")))
