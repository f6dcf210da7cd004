"""tf.app.flags-compatible flag system (reference .idea/MNISTDist.py:13-31, :37, :197-198).

Same surface as the TF 0.x wrapper the reference uses: ``DEFINE_string /
DEFINE_integer / DEFINE_float / DEFINE_boolean(name, default, help)``, a global
``FLAGS`` object parsed lazily on first attribute access, ``--name=value`` and
``--name value`` syntax, ``--[no]name`` booleans, and ``app.run(main)`` which
parses argv and calls ``main(remaining_argv)``.
"""
from __future__ import annotations

import sys
from typing import Any, Callable, Dict, List, Optional


class _Flag:
    __slots__ = ("name", "default", "help", "kind", "value", "present")

    def __init__(self, name, default, help, kind):
        self.name, self.default, self.help, self.kind = name, default, help, kind
        self.value = default
        self.present = False


def _parse_bool(s: str) -> bool:
    t = s.strip().lower()
    if t in ("1", "true", "t", "yes", "y"):
        return True
    if t in ("0", "false", "f", "no", "n"):
        return False
    raise ValueError(f"invalid boolean value {s!r}")


_CASTS: Dict[str, Callable[[str], Any]] = {
    "string": str,
    "integer": int,
    "float": float,
    "boolean": _parse_bool,
}


class FlagValues:
    """Global flag registry; attribute access triggers parsing of sys.argv once."""

    def __init__(self):
        object.__setattr__(self, "_flags", {})
        object.__setattr__(self, "_aliases", {})
        object.__setattr__(self, "_parsed", False)
        object.__setattr__(self, "_remaining", [])

    # -- definition ---------------------------------------------------------
    def _define(self, name, default, help, kind):
        if name in self._flags:
            raise ValueError(f"flag --{name} defined twice")
        self._flags[name] = _Flag(name, default, help, kind)

    def _alias(self, alias, target):
        if target not in self._flags:
            raise KeyError(target)
        self._aliases[alias] = target

    # -- parsing ------------------------------------------------------------
    def _parse(self, argv: Optional[List[str]] = None) -> List[str]:
        args = list(sys.argv[1:] if argv is None else argv)
        remaining: List[str] = []
        i = 0
        while i < len(args):
            a = args[i]
            i += 1
            if a == "--":
                remaining.extend(args[i:])
                break
            if not a.startswith("-") or a == "-":
                remaining.append(a)
                continue
            body = a.lstrip("-")
            name, eq, val = body.partition("=")
            name = self._aliases.get(name, name)
            flag = self._flags.get(name)
            if flag is None and not eq and name.startswith("no"):
                cand = self._aliases.get(name[2:], name[2:])
                f2 = self._flags.get(cand)
                if f2 is not None and f2.kind == "boolean":
                    f2.value, f2.present = False, True
                    continue
            if flag is None:
                remaining.append(a)  # unknown flags are passed through, as tf.app.flags did
                continue
            if flag.kind == "boolean" and not eq:
                # `--flag` alone means True; `--flag false` is also accepted
                if i < len(args) and args[i].lower() in ("true", "false", "0", "1"):
                    val = args[i]
                    i += 1
                else:
                    flag.value, flag.present = True, True
                    continue
            elif not eq:
                if i >= len(args):
                    raise ValueError(f"flag --{name} needs a value")
                val = args[i]
                i += 1
            try:
                flag.value = _CASTS[flag.kind](val)
            except ValueError as e:
                raise ValueError(f"bad value for --{name}: {val!r} ({e})") from None
            flag.present = True
        object.__setattr__(self, "_parsed", True)
        object.__setattr__(self, "_remaining", remaining)
        return remaining

    def _reset(self):
        for f in self._flags.values():
            f.value, f.present = f.default, False
        object.__setattr__(self, "_parsed", False)

    # -- access -------------------------------------------------------------
    def __getattr__(self, name):
        flags = object.__getattribute__(self, "_flags")
        name = object.__getattribute__(self, "_aliases").get(name, name)
        if name not in flags:
            raise AttributeError(name)
        if not object.__getattribute__(self, "_parsed"):
            self._parse()
        return flags[name].value

    def __setattr__(self, name, value):
        name = self._aliases.get(name, name)
        if name not in self._flags:
            raise AttributeError(name)
        self._flags[name].value = value
        self._flags[name].present = True

    def is_set(self, name: str) -> bool:
        return self._flags[self._aliases.get(name, name)].present

    def flag_values_dict(self) -> Dict[str, Any]:
        if not self._parsed:
            self._parse()
        return {k: f.value for k, f in self._flags.items()}

    def usage(self) -> str:
        lines = []
        for f in self._flags.values():
            lines.append(f"  --{f.name} ({f.kind}, default {f.default!r}): {f.help}")
        return "\n".join(lines)


FLAGS = FlagValues()


def DEFINE_string(name, default, help=""):
    FLAGS._define(name, default, help, "string")


def DEFINE_integer(name, default, help=""):
    FLAGS._define(name, default, help, "integer")


def DEFINE_float(name, default, help=""):
    FLAGS._define(name, default, help, "float")


def DEFINE_boolean(name, default, help=""):
    FLAGS._define(name, default, help, "boolean")


DEFINE_bool = DEFINE_boolean


def DEFINE_alias(alias, target):
    """``--alias`` is another spelling of ``--target`` (e.g. --steps for --training_iter)."""
    FLAGS._alias(alias, target)


class app:  # noqa: N801  (mirrors tf.app)
    @staticmethod
    def run(main=None, argv=None):
        rest = FLAGS._parse(None if argv is None else list(argv[1:]))
        if "--help" in rest or "-h" in rest:
            print(FLAGS.usage())
            sys.exit(0)
        main = main or sys.modules["__main__"].main
        sys.exit(main([sys.argv[0] if argv is None else argv[0]] + rest))
