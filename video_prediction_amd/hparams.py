"""Minimal re-implementation of tf.contrib.training.HParams as used by the reference.

The reference builds its hyper-parameters with ``HParams(**defaults).override_from_dict(json_dict)`` followed by
``.parse("k=v,k=[a,b]")`` (/root/reference/video_prediction/models/base_model.py:99-109).  tf.contrib is not
importable here, so the same surface is provided: values are typed by their default, unknown names raise
``ValueError``, tuple/list defaults accept ``name=[1,2]`` syntax, ``.values()`` returns a plain dict.
"""
import re


def _cast(name, default, value):
    if isinstance(default, bool):
        if isinstance(value, str):
            lv = value.strip().lower()
            if lv in ('true', '1'):
                return True
            if lv in ('false', '0'):
                return False
            raise ValueError('Could not parse hparam %s=%r as bool' % (name, value))
        return bool(value)
    if isinstance(default, int) and not isinstance(default, bool):
        if isinstance(value, float) and value != int(value):
            raise ValueError('Could not parse hparam %s=%r as int' % (name, value))
        return int(value)
    if isinstance(default, float):
        return float(value)
    if isinstance(default, str):
        return str(value)
    return value


class HParams(object):
    def __init__(self, **kwargs):
        object.__setattr__(self, '_defaults', dict(kwargs))
        object.__setattr__(self, '_values', dict(kwargs))

    def __getattr__(self, name):
        try:
            return object.__getattribute__(self, '_values')[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self.set_hparam(name, value)

    def __contains__(self, name):
        return name in self._values

    def set_hparam(self, name, value):
        if name not in self._values:
            raise ValueError('Unknown hyperparameter: %s' % name)
        default = self._defaults[name]
        if isinstance(default, (tuple, list)):
            if not isinstance(value, (tuple, list)):
                raise ValueError('Must pass a list for multi-valued parameter: %s' % name)
            elem = default[0] if len(default) else None
            vals = [(_cast(name, elem, v) if elem is not None else v) for v in value]
            self._values[name] = type(default)(vals) if isinstance(default, tuple) else vals
        else:
            if isinstance(value, (tuple, list)):
                raise ValueError('Must not pass a list for single-valued parameter: %s' % name)
            self._values[name] = _cast(name, default, value)

    def override_from_dict(self, values_dict):
        for name, value in values_dict.items():
            self.set_hparam(name, value)
        return self

    _PARAM_RE = re.compile(r'\s*(?P<name>[a-zA-Z][\w\.]*)\s*=\s*((?P<val>[^,\[]*)|\[(?P<vals>[^\]]*)\])\s*($|,)')

    def parse(self, values):
        pos = 0
        while pos < len(values):
            m = self._PARAM_RE.match(values, pos)
            if not m:
                raise ValueError('Malformed hyperparameter value: %s' % values[pos:])
            pos = m.end()
            name = m.group('name')
            if m.group('vals') is not None:
                items = [v.strip() for v in m.group('vals').split(',') if v.strip() != '']
                self.set_hparam(name, items)
            else:
                self.set_hparam(name, m.group('val').strip())
        return self

    def values(self):
        return dict(self._values)

    def get(self, name, default=None):
        return self._values.get(name, default)

    def __repr__(self):
        return 'HParams(%s)' % ', '.join('%s=%r' % kv for kv in sorted(self._values.items()))
