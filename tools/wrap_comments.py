"""developer tool: bring comment text within a column limit without touching a token of code. Pure `//` comment lines longer than the limit are re-flowed (the block's indentation and
`// ` prefix kept, continuation lines indented like the first); a trailing `// comment` that pushes a code line over the limit moves to its own line(s) above the code, at the code's
indentation. Lines that are part of a macro continuation (`\\` at the end, or following one) and code that is too long by itself are left alone. usage: wrap_comments.py [--limit N] files..."""
import re, sys, textwrap

def flow(prefix, text, limit):
    return [prefix + l for l in textwrap.wrap(text, width=max(40, limit - len(prefix)), break_long_words=False, break_on_hyphens=False)] or [prefix.rstrip()]

def split_trailing(line):
    """index of a trailing // comment outside string / char literals, or -1"""
    i, n, q = 0, len(line), None
    while i < n:
        ch = line[i]
        if q:
            if ch == '\\': i += 2; continue
            if ch == q: q = None
        elif ch in '"\'': q = ch
        elif ch == '/' and i + 1 < n and line[i + 1] == '/': return i
        i += 1
    return -1

def process(path, limit):
    L = open(path).read().split('\n'); out = []; changed = 0; k = 0
    def pure(l): s = l.lstrip(); return s.startswith('//') and not s.startswith('///')
    def cont(k): return L[k].rstrip().endswith('\\') or (k > 0 and L[k - 1].rstrip().endswith('\\'))
    def structural(body): return body.strip() == '' or re.match(r'\s*(?:\*|\d+\.|\(\w+\)|-|\|)\s', body) is not None or body.startswith('   ')
    while k < len(L):
        l = L[k]
        if pure(l) and not cont(k):
            # a paragraph: consecutive pure comment lines of one indentation, none of them structural (list items, tables, indented continuation text keep their lines)
            s = l.lstrip(); ind = l[:len(l) - len(s)]; m = re.match(r'(//\s?)(.*)', s); body = m.group(2)
            if structural(body):
                if len(l) > limit:
                    m2 = re.match(r'(\s*(?:\*|\d+\.|\(\w+\)|-)\s+)(.*)', body)
                    if m2:
                        w = flow(ind + '// ' + m2.group(1), m2.group(2), limit); hang = ind + '// ' + ' ' * len(m2.group(1))
                        out += w[:1] + [hang + r[len(ind + '// ' + m2.group(1)):] for r in w[1:]]; changed += 1; k += 1; continue
                out.append(l); k += 1; continue
            j = k; para = []
            while j < len(L) and pure(L[j]) and not cont(j) and L[j][:len(L[j]) - len(L[j].lstrip())] == ind:
                b = re.match(r'(//\s?)(.*)', L[j].lstrip()).group(2)
                if structural(b): break
                para.append(b); j += 1
            if any(len(L[q]) > limit for q in range(k, j)): out += flow(ind + '// ', ' '.join(x.strip() for x in para), limit); changed += 1
            else: out += L[k:j]
            k = j; continue
        if len(l) > limit and not cont(k):
            s = l.lstrip(); ind = l[:len(l) - len(s)]; j = split_trailing(l)
            if j > 0 and len(l[:j].rstrip()) <= limit and l[:j].strip():
                out += flow(ind + '// ', l[j + 2:].strip(), limit); out.append(l[:j].rstrip()); changed += 1; k += 1; continue
        out.append(l); k += 1
    if changed: open(path, 'w').write('\n'.join(out))
    return changed

if __name__ == '__main__':
    a = sys.argv[1:]; limit = 160
    if a and a[0] == '--limit': limit = int(a[1]); a = a[2:]
    for f in a: print(f, process(f, limit), 'lines re-flowed')
