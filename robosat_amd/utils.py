"""The training-history artifact ``history-NNNNN-of-NNNNN.png`` that ``rs train`` leaves beside every checkpoint
(reference robosat/utils.py:8-25 draws it; tools/train.py:146-148 names it): one curve per history entry over the
epochs, epochs counted from 1 on the x axis, grid, legend in the history's own key order."""

from matplotlib.backends.backend_agg import FigureCanvasAgg
from matplotlib.figure import Figure


def plot(out, history):
    """``history``: ordered mapping name -> list of per-epoch values (possibly of different lengths)."""

    names = list(history)
    epochs = max((len(history[k]) for k in names), default=0)
    figure = Figure()
    FigureCanvasAgg(figure)  # an Agg canvas of its own: no pyplot state, no display, safe in every rank / thread
    axes = figure.add_subplot(1, 1, 1)
    curves = [axes.plot(history[k])[0] for k in names]
    axes.set_xticks(range(epochs))
    axes.set_xticklabels([str(e) for e in range(1, epochs + 1)])
    axes.set_xlabel("epoch")
    axes.grid(True)
    if curves:
        axes.legend(curves, names)
    figure.savefig(out, format="png")
