#!/usr/bin/env python3
"""Re-wraps the prose of a markdown file to at most 120 columns (development tool: `python tools/reflow_md.py FILE...`).
Table rows, fenced code, headings and lines that are short enough are left alone; a list item's continuation lines are
indented to its text."""
import re
import sys
import textwrap

WIDTH = 120


def reflow(text):
    out, fence = [], False
    for line in text.split("\n"):
        if line.lstrip().startswith("```"):
            fence = not fence
            out.append(line)
            continue
        if fence or len(line) <= WIDTH or line.lstrip().startswith(("|", "#")):
            out.append(line)
            continue
        m = re.match(r"^(\s*)((?:[-*+]|\d+\.)\s+)?", line)
        indent, bullet = m.group(1), m.group(2) or ""
        body = line[len(indent) + len(bullet):]
        wrapped = textwrap.wrap(body, width=WIDTH - len(indent) - len(bullet), break_long_words=False, break_on_hyphens=False)
        for i, w in enumerate(wrapped):
            out.append(indent + (bullet if i == 0 else " " * len(bullet)) + w)
    return "\n".join(out)


if __name__ == "__main__":
    for path in sys.argv[1:]:
        with open(path) as fh:
            src = fh.read()
        with open(path, "w") as fh:
            fh.write(reflow(src))
        print(path, "max line", max(len(l) for l in reflow(src).split("\n")))
