#!/usr/bin/env python3
"""Re-flows a markdown file: joins the lines of each paragraph / list item back into one line (fenced code, tables, headings and blank lines
stay), then wraps with tools/wrap_md.py's rules.  Usage: tools/reflow_md.py <file.md> [width=120]   (in place)"""
import os
import re
import subprocess
import sys

path = sys.argv[1]
width = sys.argv[2] if len(sys.argv) > 2 else "120"
out, fence = [], False
item = re.compile(r"^\s*(?:\*|-|\d+\.)\s+")
for line in open(path, encoding="utf-8").read().split("\n"):
    if line.startswith("```"):
        fence = not fence
        out.append(line)
        continue
    if not fence and item.match(line):  # (one space behind a list marker: re-flowing must not widen the hanging indent)
        line = re.sub(r"^(\s*(?:\*|-|\d+\.))\s+", r"\1 ", line)
    special = fence or not line.strip() or line.startswith("#") or line.startswith("|") or item.match(line)
    prev_joinable = out and out[-1].strip() and not out[-1].startswith("#") and not out[-1].startswith("|") and not out[-1].startswith("```")
    if not special and prev_joinable and not fence:
        out[-1] = out[-1].rstrip() + " " + line.strip()
    else:
        out.append(line)
tmp = path + ".reflow.tmp"
open(tmp, "w", encoding="utf-8").write("\n".join(out))
subprocess.check_call([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "wrap_md.py"), tmp, path, width])
os.remove(tmp)
