"""Lists the entry points declared in include/nvalchemiops_hip.h (used by build() and the CPU-side ABI test)."""
import os
import re

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "nvalchemiops_hip.h")


def declared_symbols() -> list[str]:
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text)))


if __name__ == "__main__":
    print("\n".join(declared_symbols()))
