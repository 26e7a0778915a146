#!/usr/bin/env python3
"""Wraps the prose of a markdown file to at most WIDTH bytes per line (UTF-8: awk / wc count bytes): paragraphs and list items are re-flowed,
headings, tables and fenced code are left alone.  Usage: tools/wrap_md.py <in.md> <out.md> [width=120]"""
import re
import sys


def blen(s):
    return len(s.encode("utf-8"))


def wrap(text, width, first_indent, indent):
    words, lines, cur = text.split(), [], first_indent
    for w in words:
        cand = cur + (" " if cur.strip() else "") + w if cur.strip() else cur + w
        if blen(cand) <= width or not cur.strip():
            cur = cand
        else:
            lines.append(cur)
            cur = indent + w
    lines.append(cur)
    return lines


def main():
    src, dst = sys.argv[1], sys.argv[2]
    width = int(sys.argv[3]) if len(sys.argv) > 3 else 120
    out, fence = [], False
    for line in open(src, encoding="utf-8").read().split("\n"):
        if line.startswith("```"):
            fence = not fence
        if fence or blen(line) <= width or line.startswith("#") or line.startswith("|") or line.startswith("```"):
            out.append(line)
            continue
        m = re.match(r"^(\s*(?:\*|-|\d+\.)\s+)(.*)$", line)
        if m:
            out += wrap(m.group(2), width, m.group(1), " " * len(m.group(1)))
        else:
            out += wrap(line, width, "", "")
    open(dst, "w", encoding="utf-8").write("\n".join(out))


if __name__ == "__main__":
    main()
