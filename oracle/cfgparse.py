"""Oracle: darknet .cfg text -> list of dict blocks.  TEST INFRASTRUCTURE.

Follows reference cfg.py:198-228 (`parse_cfg`):
  * blank lines and lines starting with '#' are skipped (after rstrip)
  * '[name]' opens a block whose 'type' is name; convolutional blocks get the default
    batch_normalize = 0 (int, not str)
  * 'key = value' -> stripped strings; the key 'type' is stored as '_type'
"""


def parse_cfg(path):
    blocks = []
    cur = None
    with open(path, "r") as fh:
        for raw in fh:
            line = raw.rstrip()
            if not line or line.startswith("#"):
                continue
            if line.startswith("["):
                if cur:
                    blocks.append(cur)
                cur = {"type": line.lstrip("[").rstrip("]")}
                if cur["type"] == "convolutional":
                    cur["batch_normalize"] = 0
                continue
            key, value = line.split("=")
            key = key.strip()
            cur["_type" if key == "type" else key] = value.strip()
    if cur:
        blocks.append(cur)
    return blocks
