#!/usr/bin/env python3
"""Mechanical source transform that lets g++ compile the reference's CUDA sources
against oracle/ref_shim/include/dgemu.h (TEST INFRASTRUCTURE ONLY).

  transform.py <reference root> <output dir>

Reads dietgpu/{ans,float,utils}/*.{h,cuh,cu,cpp} where they lie, writes the
rewritten text under <output dir>/dietgpu/... (a scratch directory outside the
repository: reference sources are never copied into the repo).  Two rewrites,
both purely syntactic, nothing else is touched:

  1. kernel launches   NAME<T...><<<grid, block, smem, stream>>>(args)
       -> dgemu::launch(dgemu::LaunchCfg(grid, block, smem, stream),
                        [&](auto&&... dgemu_a) { NAME<T...>(dgemu_a...); }, args)
  2. PTX inline asm    asm("op %0, %1, ...;" : "=r"(out) : "r"(a), "r"(b) ...);
       -> dgemu::ptx_exec("op %0, %1, ...;", out, a, b, ...);
     (the opcodes are interpreted by dgemu::ptx_eval in emu.cpp)
"""
import os
import re
import sys


def skip_ws_back(s, i):
    """index of the last non-blank character before i (backslash-newlines count as blanks)"""
    while i > 0 and (s[i - 1] in " \t\r\n" or (s[i - 1] == "\\" and i < len(s) and s[i] in "\r\n")):
        i -= 1
    return i


def match_back(s, i, open_c, close_c):
    """s[i-1] == close_c: returns the index of the matching open_c"""
    depth = 0
    j = i - 1
    while j >= 0:
        if s[j] == close_c:
            depth += 1
        elif s[j] == open_c:
            depth -= 1
            if depth == 0:
                return j
        j -= 1
    raise ValueError("unbalanced " + open_c + close_c)


def match_fwd(s, i, open_c, close_c):
    """s[i] == open_c: returns the index of the matching close_c"""
    depth = 0
    j = i
    while j < len(s):
        if s[j] == open_c:
            depth += 1
        elif s[j] == close_c:
            depth -= 1
            if depth == 0:
                return j
        j += 1
    raise ValueError("unbalanced " + open_c + close_c)


def rewrite_launches(s):
    out = []
    pos = 0
    n = 0
    while True:
        k = s.find("<<<", pos)
        if k < 0:
            break
        close = s.find(">>>", k)
        assert close > 0
        cfg = s[k + 3 : close]
        # kernel expression: identifier [<template args>] right before <<<
        e = skip_ws_back(s, k)
        start = e
        if s[start - 1] == ">":
            start = match_back(s, start, "<", ">")
            start = skip_ws_back(s, start)
        while start > 0 and (s[start - 1].isalnum() or s[start - 1] in "_:"):
            start -= 1
        name = s[start:e]
        assert re.match(r"[A-Za-z_]", name), (name, s[k - 80 : k + 20])
        # argument list
        a0 = close + 3
        while s[a0] in " \t\r\n\\":
            a0 += 1
        assert s[a0] == "(", s[close : close + 20]
        a1 = match_fwd(s, a0, "(", ")")
        args = s[a0 + 1 : a1]
        sep = ", " if args.strip() else ""
        out.append(s[pos:start])
        out.append(
            "dgemu::launch(dgemu::LaunchCfg(%s), [&](auto&&... dgemu_a) { %s(dgemu_a...); }%s%s)"
            % (cfg, name, sep, args)
        )
        pos = a1 + 1
        n += 1
    out.append(s[pos:])
    return "".join(out), n


ASM_RE = re.compile(r'\basm\s*(?:volatile\s*)?\(\s*("(?:[^"\\]|\\.)*")\s*:', re.S)


def split_operands(text):
    """'"r"(a), "r"((uint32_t)b)' -> ['a', '(uint32_t)b']"""
    ops = []
    i = 0
    while True:
        m = re.compile(r'\s*"[^"]*"\s*\(', re.S).match(text, i)
        if not m:
            break
        p0 = m.end() - 1
        p1 = match_fwd(text, p0, "(", ")")
        ops.append(text[p0 + 1 : p1].strip())
        i = p1 + 1
        m2 = re.compile(r"\s*,", re.S).match(text, i)
        if m2:
            i = m2.end()
    return ops


def rewrite_asm(s):
    out = []
    pos = 0
    n = 0
    for m in ASM_RE.finditer(s):
        line_start = s.rfind("\n", 0, m.start()) + 1
        if s[line_start : m.start()].lstrip().startswith("//"):
            continue  # commented-out asm
        p0 = s.index("(", m.start())
        p1 = match_fwd(s, p0, "(", ")")
        body = s[m.end() : p1]  # after the first ':'
        parts = body.split(":")
        outs = split_operands(parts[0])
        ins = split_operands(parts[1]) if len(parts) > 1 else []
        assert len(outs) == 1, s[m.start() : p1]
        end = p1 + 1
        while s[end] in " \t\r\n":
            end += 1
        assert s[end] == ";"
        out.append(s[pos : m.start()])
        out.append("dgemu::ptx_exec(%s, %s%s);" % (m.group(1), outs[0], "".join(", " + x for x in ins)))
        pos = end + 1
        n += 1
    out.append(s[pos:])
    return "".join(out), n


def main():
    ref, dst = sys.argv[1], sys.argv[2]
    total_l = total_a = 0
    for sub in ("ans", "float", "utils"):
        src_dir = os.path.join(ref, "dietgpu", sub)
        for f in sorted(os.listdir(src_dir)):
            if not f.endswith((".h", ".cuh", ".cu", ".cpp")) or "Test" in f:
                continue
            text = open(os.path.join(src_dir, f)).read()
            text, nl = rewrite_launches(text)
            text, na = rewrite_asm(text)
            total_l += nl
            total_a += na
            out_dir = os.path.join(dst, "dietgpu", sub)
            os.makedirs(out_dir, exist_ok=True)
            name = f + ".cpp" if f.endswith(".cu") else f
            with open(os.path.join(out_dir, name), "w") as o:
                o.write(text)
    print("transform.py: %d kernel launches, %d asm statements rewritten" % (total_l, total_a))


if __name__ == "__main__":
    main()
