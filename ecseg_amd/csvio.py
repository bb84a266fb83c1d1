"""CSV text exactly as ``pandas.DataFrame.to_csv(index=False)`` writes it for the two tasks (reference
src/metaseg.py:56-57, src/meta_overlay.py:98-102)."""

METASEG_COLUMNS = ['image name', '# of ec']
OVERLAY_COLUMNS = ['image_name', '# of ecDNA (DAPI)', '# of ecDNA (green)', '# of ecDNA (red)',
                   '# of ecDNA (DAPI and green)', '# of ecDNA (DAPI and red)', '# of ecDNA (red and green)',
                   '# of ecDNA (DAPI and red and green)', '# of HSR (red)', '# of HSR (green)']


def _cell(v):
    if isinstance(v, tuple):                    # count_cc tuples: str((n, px)) with px float 0.0 or an int
        n, px = v
        s = '(%d, %s)' % (n, '0.0' if isinstance(px, float) else '%d' % px)
    else:
        s = str(v)
    if any(ch in s for ch in ',"\n\r'):
        s = '"%s"' % s.replace('"', '""')
    return s


def csv_text(columns, rows):
    lines = [','.join(_cell(c) for c in columns)]
    lines += [','.join(_cell(v) for v in r) for r in rows]
    return '\n'.join(lines) + '\n'


def overlay_cells(rec12):
    """12 int64 fields of ecseg_overlay -> the nine CSV cells (three (n, px) tuples + six ints)."""
    cells = []
    for j in (0, 2, 4):
        n, px = int(rec12[j]), int(rec12[j + 1])
        cells.append((n, 0.0 if px == -1 else px))
    return cells + [int(v) for v in rec12[6:12]]
