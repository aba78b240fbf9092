#!/usr/bin/env python3
"""INTEGRATION.md quotes the binding from the file that is compiled and tested (oracle/ref_host_gpu.c): this copies the text between the
file's two section markers into the document's `<!-- binding:begin/end -->` block.  `--check`: exit 1 if they differ (tests)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def binding_text():
    src = open(os.path.join(ROOT, "oracle", "ref_host_gpu.c")).read()
    a = src.index("/* ===================== the binding")
    b = src.index("/* ===================== what sim_main does")
    return "```c\n" + src[a:b].rstrip() + "\n```\n"


def main():
    path = os.path.join(ROOT, "INTEGRATION.md")
    doc = open(path).read()
    a = doc.index("<!-- binding:begin -->") + len("<!-- binding:begin -->\n")
    b = doc.index("<!-- binding:end -->")
    new = doc[:a] + binding_text() + doc[b:]
    if "--check" in sys.argv:
        sys.exit(0 if new == doc else 1)
    open(path, "w").write(new)


if __name__ == "__main__":
    main()
