"""Training-history plot (reference ``robosat/utils.py``)."""

import matplotlib

matplotlib.use("Agg")
import matplotlib.pyplot as plt  # noqa: E402


def plot(out, history):
    plt.figure()
    n = max(len(v) for v in history.values())
    plt.xticks(range(n), [i + 1 for i in range(n)])
    plt.grid()
    for values in history.values():
        plt.plot(values)
    plt.xlabel("epoch")
    plt.legend(list(history))
    plt.savefig(out, format="png")
    plt.close()
