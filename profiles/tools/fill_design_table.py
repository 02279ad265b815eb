#!/usr/bin/env python3
"""Put the table of profiles/tools/roofline_table.py between the roofline-table markers of DESIGN.md:  fill_design_table.py profiles/r03"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
table = subprocess.run([sys.executable, os.path.join(root, "profiles", "tools", "roofline_table.py"), sys.argv[1]],
                       capture_output=True, text=True, check=True).stdout
p = os.path.join(root, "DESIGN.md")
s = open(p).read()
s = re.sub(r"<!-- roofline-table:begin -->.*?<!-- roofline-table:end -->",
           lambda m: "<!-- roofline-table:begin -->\n" + table + "<!-- roofline-table:end -->", s, flags=re.S)
open(p, "w").write(s)
