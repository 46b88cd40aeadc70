#!/usr/bin/env python
"""Resolve a given set of preprocessor switches as UNDEFINED in a source file, leaving every other conditional alone (a tiny `unifdef -U`).
Handles `#ifdef X`, `#ifndef X`, `#if defined(X)`, `#if !defined(X)`, `#else`, `#endif`, nested.  usage: unifdef_lite.py <prefix-regex> <in> <out>"""
import re
import sys

pat, src, dst = re.compile(sys.argv[1]), sys.argv[2], sys.argv[3]
out, stack = [], []      # stack entries: None (foreign conditional, kept verbatim) or [emitting_now, seen_else]
for line in open(src):
    st = line.strip()
    m = re.match(r"#\s*(ifdef|ifndef)\s+(\w+)", st) or re.match(r"#\s*if\s+(!?)\s*defined\s*\(\s*(\w+)\s*\)\s*(?://.*|/\*.*)?$", st)
    emitting = all(e is None or e[0] for e in stack)
    if m and pat.fullmatch(m.group(2)):
        neg = m.group(1) in ("ifndef", "!")
        stack.append([neg, False])            # macro undefined: `ifndef` / `!defined` bodies are kept
        continue
    if re.match(r"#\s*if", st):
        stack.append(None)
        if emitting:
            out.append(line)
        continue
    if re.match(r"#\s*else\b", st) and stack and stack[-1] is not None:
        stack[-1][0] = not stack[-1][0]
        continue
    if re.match(r"#\s*endif\b", st) and stack:
        e = stack.pop()
        if e is None and all(x is None or x[0] for x in stack):
            out.append(line)
        continue
    if emitting:
        out.append(line)
assert not stack, "unbalanced conditionals"
open(dst, "w").write("".join(out))
