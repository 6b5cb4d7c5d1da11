#!/usr/bin/env python3
"""Re-wraps the prose of a markdown file to at most 120 columns (development tool: `python tools/reflow_md.py FILE...`).
Paragraphs and list items that hold a line longer than that are joined and wrapped again as a whole (a list item's
continuation lines are indented to its text); table rows, fenced code, headings and blocks that already fit are left alone
(`--all`: every paragraph is wrapped again)."""
import re
import sys
import textwrap

WIDTH = 120
BULLET = re.compile(r"^(\s*)((?:[-*+]|\d+\.)\s+)")


def reflow(text, force=False):
    lines = text.split("\n")
    out, i, fence = [], 0, False

    def special(l):
        return (not l.strip()) or l.lstrip().startswith(("|", "#", "```", ">")) or l.strip() in ("---", "***")

    while i < len(lines):
        l = lines[i]
        if l.lstrip().startswith("```"):
            fence = not fence
            out.append(l)
            i += 1
            continue
        if fence or special(l):
            out.append(l)
            i += 1
            continue
        # a block: this line and the following lines that continue it (not blank, not special, not a new list item)
        j = i + 1
        while j < len(lines) and not special(lines[j]) and not BULLET.match(lines[j]):
            j += 1
        block = lines[i:j]
        if not force and max(len(b) for b in block) <= WIDTH:
            out += block
        else:
            m = BULLET.match(block[0])
            indent = m.group(1) if m else re.match(r"^\s*", block[0]).group(0)
            bullet = m.group(2) if m else ""
            body = " ".join([block[0][len(indent) + len(bullet):].strip()] + [b.strip() for b in block[1:]])
            wrapped = textwrap.wrap(body, width=WIDTH - len(indent) - len(bullet), break_long_words=False, break_on_hyphens=False)
            for k, w in enumerate(wrapped):
                out.append(indent + (bullet if k == 0 else " " * len(bullet)) + w)
        i = j
    return "\n".join(out)


if __name__ == "__main__":
    force = "--all" in sys.argv  # re-wrap every paragraph, also those that already fit
    for path in [a for a in sys.argv[1:] if a != "--all"]:
        with open(path) as fh:
            src = fh.read()
        res = reflow(src, force)
        with open(path, "w") as fh:
            fh.write(res)
        print(path, "max line", max(len(l) for l in res.split("\n")))
