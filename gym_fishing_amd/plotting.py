"""Plots of simulate() / policyfn() tables -- the reference's env.plot / env.plot_policy
(base_fishing_env.py:103-110 -> shared_env.py:105-127).  Host-side convenience only;
matplotlib is imported lazily with the non-interactive Agg backend."""
import numpy as np


def _pyplot():
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    return plt


def _col(df, name):
    return np.asarray(df[name])


def plot_mdp(df, output="results.png"):
    """Three stacked panels per rep: stock, quota, cumulative reward against time."""
    plt = _pyplot()
    rep, time = _col(df, "rep"), _col(df, "time")
    panels = (("state", _col(df, "state"), False), ("action", _col(df, "action"), False),
              ("reward", _col(df, "reward"), True))
    fig, axes = plt.subplots(len(panels), 1, sharex=True)
    for r in np.unique(rep):
        sel = rep == r
        for ax, (_, values, cumulative) in zip(axes, panels):
            y = np.cumsum(values[sel]) if cumulative else values[sel]
            ax.plot(time[sel], y, color="tab:blue", alpha=0.3)
    for ax, (label, _, _) in zip(axes, panels):
        ax.set_ylabel(label)
    fig.tight_layout()
    fig.savefig(output)
    plt.close(fig)
    return output


def plot_policyfn(df, output="policy.png"):
    """Escapement (state minus quota) against state, one line per rep."""
    plt = _pyplot()
    rep, state, action = _col(df, "rep"), _col(df, "state"), _col(df, "action")
    fig, ax = plt.subplots()
    for r in np.unique(rep):
        sel = rep == r
        ax.plot(state[sel], state[sel] - action[sel], color="tab:blue")
    ax.set_xlabel("state")
    ax.set_ylabel("escapement")
    fig.savefig(output)
    plt.close(fig)
    return output
