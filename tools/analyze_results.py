#!/usr/bin/env python3
"""analyze_results.py <original.bmp> <compressed.jpg> -- size and quality report for one pair
(the reference's analysis step without the plot; metrics in jpegamd/quality.py)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "jpeg-image-compression_amd" / "python"))
from jpegamd import quality  # noqa: E402

if len(sys.argv) != 3:
    print(__doc__)
    sys.exit(1)
for p in sys.argv[1:]:
    if not Path(p).exists():
        print(f"Error: The file '{p}' was not found.")
        sys.exit(1)
print(quality.format_report(quality.analyze(Path(sys.argv[1]).read_bytes(), Path(sys.argv[2]).read_bytes())))
