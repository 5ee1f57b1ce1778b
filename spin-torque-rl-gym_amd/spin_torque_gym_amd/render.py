"""Host-side rendering of the N = 1 facades (matplotlib, optional): what `render()` of the reference's envs shows.

  macrospin env (envs/spin_torque_env.py:556-684)
      'human'     : one persistent 12 x 8 in figure, four panels -- the magnetisation and its target as 3-D arrows inside a
                    translucent unit sphere; per-step energy; alignment with the success threshold; applied current -- the
                    three histories from `episode_history`; redrawn in place on every call
      'rgb_array' : the x-y projection of both arrows in the unit circle, returned as uint8 [H, W, 3]
  array env (envs/array_env.py:594-708)
      'human'     : m_z maps of the current and the target pattern, similarity with its threshold, energy per step
      'rgb_array' : the two m_z maps side by side

Nothing here touches the GPU: the facades keep the numbers (state pulled after every step, the history list) on the host, as the
reference does.  The drawing is data-driven (`_Series`, `_Panel` below) instead of one long procedure per mode, and it is safe
under a non-interactive backend such as Agg: a figure is only `show`n / paused when the backend is interactive, and
`frame_of(fig)` reads the canvas through `buffer_rgba()` (`tostring_rgb`, which the reference calls, no longer exists in current
matplotlib).  Without matplotlib `render()` warns and returns None, like the reference.
"""
import warnings
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional, Sequence

import numpy as np


def _pyplot():
    try:
        import matplotlib
        import matplotlib.pyplot as plt
        return matplotlib, plt
    except ImportError:
        warnings.warn("Matplotlib not available, rendering disabled")
        return None, None


def frame_of(fig) -> np.ndarray:
    """The figure's canvas as uint8 [H, W, 3]."""
    fig.canvas.draw()
    return np.asarray(fig.canvas.buffer_rgba())[..., :3].copy()


@dataclass
class _Series:
    """One history curve of a 2-D panel: key of the `episode_history` entries (or a callable on an entry), line style."""
    value: Any
    style: str
    label: Optional[str] = None

    def of(self, entry: Dict[str, Any]):
        return self.value(entry) if callable(self.value) else entry[self.value]


@dataclass
class _Panel:
    title: str
    xlabel: str
    ylabel: str
    series: Sequence[_Series]
    hline: Optional[Callable[[Any], float]] = None       # env -> y of a dashed threshold line
    hline_label: str = ""
    ylim: Optional[Sequence[float]] = None

    def draw(self, ax, env):
        hist = env.episode_history
        if not hist:
            return
        steps = [h["step"] for h in hist]
        for s in self.series:
            ax.plot(steps, [s.of(h) for h in hist], s.style, **({"label": s.label} if s.label else {}))
        if self.hline is not None:
            ax.axhline(y=self.hline(env), color="r", linestyle="--", label=self.hline_label)
        ax.set_xlabel(self.xlabel)
        ax.set_ylabel(self.ylabel)
        ax.set_title(self.title)
        if self.ylim is not None:
            ax.set_ylim(list(self.ylim))
        if self.hline is not None or any(s.label for s in self.series):
            ax.legend()


MACROSPIN_PANELS = (
    _Panel("Energy Consumption", "Step", "Energy (J)", [_Series("energy", "g-")]),
    _Panel("Target Alignment", "Step", "Alignment", [_Series("alignment", "b-")],
           hline=lambda env: env.success_threshold, hline_label="Success threshold"),
    _Panel("Applied Current", "Step", "Current (A/m²)", [_Series(lambda h: h["action"][0], "orange")]),
)
ARRAY_PANELS = (
    _Panel("Pattern Similarity Progress", "Step", "Pattern Similarity", [_Series("similarity", "b-", "Similarity")],
           hline=lambda env: env.success_threshold, hline_label="Success threshold", ylim=(0, 1)),
    _Panel("Energy Consumption per Step", "Step", "Energy (J)", [_Series("energy", "g-")]),
)


def _sphere(n=50):
    u, v = np.linspace(0, 2 * np.pi, n), np.linspace(0, np.pi, n)
    return np.outer(np.cos(u), np.sin(v)), np.outer(np.sin(u), np.sin(v)), np.outer(np.ones(n), np.cos(v))


def _arrows3d(ax, env):
    for vec, colour, name in ((env.current_magnetization, "red", "Current"), (env.target_magnetization, "blue", "Target")):
        ax.quiver(0, 0, 0, *[float(x) for x in vec], color=colour, label=name, arrow_length_ratio=0.1)
    ax.plot_surface(*_sphere(), alpha=0.1, color="gray")
    for setter in (ax.set_xlim, ax.set_ylim, ax.set_zlim):
        setter([-1.5, 1.5])
    ax.set_xlabel("X"); ax.set_ylabel("Y"); ax.set_zlabel("Z")
    ax.legend()
    ax.set_title("Magnetization State")


def _mz_map(plt, fig, ax, pattern, title, labels=True):
    im = ax.imshow(np.asarray(pattern)[:, :, 2], cmap="RdBu", vmin=-1, vmax=1)
    ax.set_title(title)
    if labels:
        ax.set_xlabel("Column"); ax.set_ylabel("Row")
    return fig.colorbar(im, ax=ax)


class HumanFigure:
    """The persistent figure of render('human'): created once (when an env is built with render_mode='human', or at the first
    such call), cleared and redrawn on every call, closed by env.close().  `kind`: 'macrospin' | 'array'."""

    def __init__(self, kind: str):
        self.kind = kind
        self.fig = None
        self.axes: List[Any] = []
        self._bars: List[Any] = []
        self.ok = False
        _, plt = _pyplot()
        if plt is None:
            return
        self.fig = plt.figure(figsize=(12, 8))
        if kind == "macrospin":
            self.axes = [self.fig.add_subplot(221, projection="3d")] + [self.fig.add_subplot(220 + k) for k in (2, 3, 4)]
        else:
            self.axes = [self.fig.add_subplot(220 + k) for k in (1, 2, 3, 4)]
        self.ok = True

    def draw(self, env) -> None:
        if not self.ok:
            return
        matplotlib, plt = _pyplot()
        try:
            for cb in self._bars:                       # (a colour bar owns an axes of its own: remove it before redrawing)
                cb.remove()
            self._bars = []
            for ax in self.axes:
                ax.clear()
            if self.kind == "macrospin":
                _arrows3d(self.axes[0], env)
                for ax, panel in zip(self.axes[1:], MACROSPIN_PANELS):
                    panel.draw(ax, env)
            else:
                self._bars.append(_mz_map(plt, self.fig, self.axes[0], env.current_pattern, "Current Pattern (Mz)"))
                self._bars.append(_mz_map(plt, self.fig, self.axes[1], env.target_pattern, "Target Pattern (Mz)"))
                for ax, panel in zip(self.axes[2:], ARRAY_PANELS):
                    panel.draw(ax, env)
            self.fig.tight_layout()
            self.fig.canvas.draw()
            if matplotlib.get_backend().lower() not in ("agg", "pdf", "svg", "ps", "cairo", "template") and plt.isinteractive():
                plt.pause(0.01)
        except Exception as e:  # noqa: BLE001 -- spin_torque_env.py:653-655: a drawing problem is a warning, never an env error
            warnings.warn(f"Rendering error: {e}")

    def close(self) -> None:
        if self.fig is not None:
            _, plt = _pyplot()
            if plt is not None:
                plt.close(self.fig)
        self.fig, self.axes, self._bars, self.ok = None, [], [], False


def macrospin_rgb(env) -> Optional[np.ndarray]:
    _, plt = _pyplot()
    if plt is None:
        return None
    fig, ax = plt.subplots(figsize=(8, 6))
    m, t = env.current_magnetization, env.target_magnetization
    for vec, colour, name in ((m, "red", "Current"), (t, "blue", "Target")):
        ax.quiver(0, 0, float(vec[0]), float(vec[1]), color=colour, scale=1, label=name)
    ax.add_patch(plt.Circle((0, 0), 1, fill=False, color="gray", alpha=0.5))
    ax.set_xlim([-1.5, 1.5]); ax.set_ylim([-1.5, 1.5])
    ax.set_aspect("equal")
    ax.legend()
    ax.set_title(f"Step {env.step_count}: Alignment = {float(np.dot(m, t)):.3f}")
    rgb = frame_of(fig)
    plt.close(fig)
    return rgb


def array_rgb(env, similarity: float) -> Optional[np.ndarray]:
    _, plt = _pyplot()
    if plt is None:
        return None
    fig, (ax1, ax2) = plt.subplots(1, 2, figsize=(10, 4))
    _mz_map(plt, fig, ax1, env.current_pattern, "Current Pattern", labels=False)
    _mz_map(plt, fig, ax2, env.target_pattern, "Target Pattern", labels=False)
    fig.suptitle(f"Step {env.step_count}: Similarity = {similarity:.3f}")
    rgb = frame_of(fig)
    plt.close(fig)
    return rgb
