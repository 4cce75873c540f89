"""Re-flow a markdown file to a maximum line width (default 160 columns) without changing a word of it.
    python tools/wrap_md.py DESIGN.md HISTORY.md [--width 160] [--check]
Paragraphs and list items are re-wrapped (continuation lines of a list item are indented to its text); fenced code blocks, headings and tables whose rows
already fit are left alone.  A table with a row wider than the limit cannot be wrapped as a table: it becomes a list, one item per row --
`* **<first header>: <first cell>**` and below it one sub-item `<header>: <cell>` per further column -- same cells, same order.  `--check` only reports
(exit code 1 if a line outside a code block is wider than the limit).  The words of the file (whitespace-separated tokens outside tables) are compared
before and after; a difference aborts without writing."""
import re
import sys

BULLET = re.compile(r"^(\s*)([*+-]|\d+[.)])\s+")


def cells(line):
    """Cells of a table row; a `|` inside a code span or escaped as `\\|` does not split."""
    out, cur, tick, k = [], [], False, 0
    s = line.strip()
    s = s[1:] if s.startswith("|") else s
    while k < len(s):
        c = s[k]
        if c == "`":
            tick = not tick
        if c == "\\" and k + 1 < len(s) and s[k + 1] == "|":
            cur.append("\\|"); k += 2
            continue
        if c == "|" and not tick:
            out.append("".join(cur).strip()); cur = []
        else:
            cur.append(c)
        k += 1
    if "".join(cur).strip():
        out.append("".join(cur).strip())
    return out


def wrap(text, width, first="", rest=""):
    """Greedy fill; widths are measured in BYTES of the UTF-8 text (what `wc -L` under the C locale and most column checks count)."""
    out, cur, empty = [], first, True
    for w in text.split():
        cand = cur + w if empty else cur + " " + w
        if len(cand.encode()) > width and not empty:
            out.append(cur); cur = rest + w
        else:
            cur = cand
        empty = False
    out.append(cur.rstrip())
    return out


def table_to_list(rows, width):
    head = cells(rows[0])
    out = []
    for r in rows[2:]:
        c = cells(r)
        if len(c) != len(head):
            raise SystemExit(f"wrap_md: a table row has {len(c)} cells under {len(head)} headers:\n{r[:200]}")
        lead = c[0] if c[0].startswith("**") or not c[0] else f"**{c[0]}**"
        out += wrap(f"{lead}" if not head[0] else f"{lead}" + ("" if head[0].lower() in ("row", "kernel", "quantity", "piece", "phase") else f" ({head[0]})"), width, "* ", "  ")
        for h, v in zip(head[1:], c[1:]):
            if v and v != "—":
                out += wrap(f"{h}: {v}" if h else v, width, "  * ", "    ")
    return out


def reflow(lines, width):
    out, k, n, table_tokens = [], 0, len(lines), []
    while k < n:
        l = lines[k]
        if l.lstrip().startswith("```"):
            out.append(l); k += 1
            while k < n and not lines[k].lstrip().startswith("```"):
                out.append(lines[k]); k += 1
            if k < n:
                out.append(lines[k]); k += 1
        elif l.startswith("|"):
            j = k
            while j < n and lines[j].startswith("|"):
                j += 1
            rows = lines[k:j]
            if max(len(r) for r in rows) <= width:
                out += rows
            else:
                out += table_to_list(rows, width)
            table_tokens.append((k, j))
            k = j
        elif not l.strip() or l.startswith("#"):
            out.append(l); k += 1
        else:
            m = BULLET.match(l)
            first = m.group(0) if m else ""
            rest = " " * len(first) if m else ""
            j, buf = k + 1, [l[len(first):]]
            while j < n and lines[j].strip() and not lines[j].startswith(("#", "|")) and not lines[j].lstrip().startswith("```") and not BULLET.match(lines[j]):
                buf.append(lines[j].strip()); j += 1
            out += wrap(" ".join(buf), width, first, rest)
            k = j
    return out, table_tokens


def words(lines, skip):
    toks = []
    for k, l in enumerate(lines):
        if not any(a <= k < b for a, b in skip):
            toks += l.split()
    return toks


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    width = int(sys.argv[sys.argv.index("--width") + 1]) if "--width" in sys.argv else 160
    if "--width" in sys.argv:
        args.remove(str(width))
    bad = 0
    for path in args:
        lines = open(path).read().split("\n")
        if "--check" in sys.argv:
            code = False
            for k, l in enumerate(lines):
                if l.lstrip().startswith("```"):
                    code = not code
                elif not code and len(l.encode()) > width:
                    print(f"{path}:{k + 1}: {len(l.encode())} bytes"); bad += 1
            continue
        new, tables = reflow(lines, width)
        kept = [t for t in tables if max(len(r) for r in lines[t[0]:t[1]]) <= width]
        converted = [t for t in tables if t not in kept]
        if not converted and words(lines, []) != words(new, []):
            raise SystemExit(f"wrap_md: {path}: the re-flowed text differs in its words; nothing written")
        if converted:  # compare everything outside the converted tables
            a = words(lines, converted)
            # the converted tables' words cannot be told apart from the rest in the output: compare multisets of the untouched part instead
            from collections import Counter
            ca, cb = Counter(a), Counter(words(new, []))
            missing = {w: c - cb.get(w, 0) for w, c in ca.items() if c > cb.get(w, 0)}
            if missing:
                raise SystemExit(f"wrap_md: {path}: words lost outside the tables: {list(missing.items())[:10]}; nothing written")
        open(path, "w").write("\n".join(new))
        print(f"{path}: {len(lines)} -> {len(new)} lines, {len(converted)} table(s) turned into lists, widest line now {max(len(l.encode()) for l in new)}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
