"""TOML configuration files (model / dataset), same keys as the reference (``robosat/config.py``, ``config/*.toml``).

Model:   [common] cuda, batch_size, image_size, checkpoint   [opt] epochs, lr, loss
Dataset: [common] dataset, classes, colors                   [weights] values
``cuda = true`` means "use the MI355X" (the HIP device shows up as torch's ``cuda``); ``cuda = false`` is rejected by
the tools because this implementation has no CPU compute path.
"""

import tomli

# Class counts the tools accept, stated once.  The reference's model takes any count but its tools are binary
# (predict.py:98 asserts two classes; metrics.py:27-41 counts a 2x2 table): here the kernels behind the losses, the C x C
# confusion matrix and the head take 2..8 classes, and a probability PNG holds one byte per non-background class in at most
# 4 channels (mode P / LA / RGB / RGBA), i.e. up to 5 classes for `rs predict` -> `rs masks`.
MIN_CLASSES, MAX_CLASSES_TRAIN, MAX_CLASSES_PREDICT = 2, 8, 5


def check_num_classes(num_classes, tool):
    """Raises ``ValueError`` when ``tool`` ("train" | "predict" | "serve") cannot handle ``num_classes`` classes."""

    hi = MAX_CLASSES_PREDICT if tool == "predict" else MAX_CLASSES_TRAIN
    if not MIN_CLASSES <= num_classes <= hi:
        raise ValueError("rs {} handles {}..{} classes; the dataset config lists {}".format(tool, MIN_CLASSES, hi, num_classes))


def load_config(path):
    """Parses the TOML file at ``path`` into a dictionary."""

    with open(path, "rb") as fp:
        return tomli.load(fp)


def _fmt(v):
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, (int, float)):
        return repr(v)
    if isinstance(v, str):
        return "'{}'".format(v) if "'" not in v else '"{}"'.format(v.replace("\\", "\\\\").replace('"', '\\"'))
    if isinstance(v, (list, tuple)):
        return "[" + ", ".join(_fmt(x) for x in v) + "]"
    raise TypeError("unsupported TOML value: {!r}".format(v))


def save_config(attrs, path):
    """Writes a two-level configuration dictionary (``{table: {key: scalar | list}}``) as TOML."""

    with open(path, "w") as fp:
        for table, entries in attrs.items():
            fp.write("[{}]\n".format(table))
            for key, value in entries.items():
                fp.write("  {} = {}\n".format(key, _fmt(value)))
            fp.write("\n")
