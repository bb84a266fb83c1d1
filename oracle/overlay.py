"""Oracle: ``meta_overlay`` row of nine counts and the CSV text of both tasks.
TEST INFRASTRUCTURE ONLY.

Restates src/meta_overlay.py:59-102, src/image_tools.py:136-146 and src/metaseg.py:39-57.
PINNED by tests/golden/overlay_rows.json / csv_text.json (outputs of the reference's counting functions and of
pandas ``to_csv`` captured in the build container).
"""
import numpy as np

from . import postproc
from .preprocess import u16_to_u8

OVERLAY_COLUMNS = ['image_name', '# of ecDNA (DAPI)', '# of ecDNA (green)', '# of ecDNA (red)',
                   '# of ecDNA (DAPI and green)', '# of ecDNA (DAPI and red)', '# of ecDNA (red and green)',
                   '# of ecDNA (DAPI and red and green)', '# of HSR (red)', '# of HSR (green)']   # meta_overlay.py:98-100
METASEG_COLUMNS = ['image name', '# of ec']                                                        # metaseg.py:39


def split_fish(rgb, sensitivity):
    """src/image_tools.py:136-146 without the PNG side effects -> (red, green) boolean masks."""
    rgb = u16_to_u8(rgb)
    return rgb[..., 0] > sensitivity, rgb[..., 1] > sensitivity


def overlay_row(labels, rgb, sensitivity):
    """The nine counts in final column order; the three ``count_cc`` cells stay (n, pixels) tuples
    (src/meta_overlay.py:70,71,79)."""
    red, green = split_fish(rgb, sensitivity)
    nuclei, chrom, ec = labels == 1, labels == 2, labels == 3
    fish = green & ~nuclei
    fish2 = red & ~nuclei
    return [
        postproc.count_cc(ec),
        postproc.count_cc(fish & ~chrom),
        postproc.count_cc(fish2 & ~chrom),
        postproc.count_colocalization(ec, fish),
        postproc.count_colocalization(ec, fish2),
        postproc.count_colocalization(fish & ~chrom, fish2 & ~chrom),
        postproc.count_colocalization(ec, fish2 & fish),
        postproc.count_HSR(chrom, fish2, postproc.HSR_SIZE_THRESHOLD),
        postproc.count_HSR(chrom, fish, postproc.HSR_SIZE_THRESHOLD),
    ]


def _cell(v):
    if isinstance(v, tuple):
        n, px = v
        s = '(%d, %s)' % (n, '0.0' if isinstance(px, float) else '%d' % px)
        return '"%s"' % s                      # contains a comma -> csv.QUOTE_MINIMAL quotes it
    s = str(v)
    if any(ch in s for ch in ',"\n\r'):
        s = '"%s"' % s.replace('"', '""')
    return s


def csv_text(columns, rows):
    """``DataFrame.to_csv(index=False)``: header line then one line per row, '\\n' terminated."""
    lines = [','.join(_cell(c) for c in columns)]
    lines += [','.join(_cell(v) for v in r) for r in rows]
    return '\n'.join(lines) + '\n'
