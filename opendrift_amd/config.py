"""Configuration mechanism with the reference's semantics (opendrift/config.py:11-212):
`key -> {type, default, min, max, enum, level, description}`, set_config validates type / range
/ enum and raises ValueError; get_config returns the value.  Host-side only."""

CONFIG_LEVEL_ESSENTIAL, CONFIG_LEVEL_BASIC, CONFIG_LEVEL_ADVANCED = 1, 2, 3


class Configurable:
    def __init__(self):
        self._config = {}

    def _add_config(self, config, overwrite=False):
        for key, item in config.items():
            if key in self._config and not overwrite:
                raise ValueError('Config item %s is already specified' % key)
            if item['type'] not in ('bool', 'float', 'int', 'enum', 'str'):
                raise ValueError('Config type "%s" (%s) is not defined' % (item['type'], key))
            item = dict(item)
            item.setdefault('default', None)
            item['value'] = item['default']
            self._config[key] = item

    def _set_config_default(self, key, value):
        self._config[key]['default'] = value
        self.set_config(key, value)

    def get_config(self, key, default=None):
        if key not in self._config:
            if default is not None:
                return default
            raise ValueError('No config setting named %s' % key)
        return self._config[key]['value']

    def get_configspec(self, prefix='', level=None):
        return {k: v for k, v in self._config.items() if k.startswith(prefix)
                and (level is None or v.get('level', 3) in level)}

    def set_config(self, key, value):
        if key not in self._config:
            import difflib
            sim = difflib.get_close_matches(key, list(self._config), n=4, cutoff=.5)
            raise ValueError('No config setting named "%s"%s' % (key, ('\nDid you mean: ' + ', '.join(sim)) if sim else ''))
        i = self._config[key]
        if value is not None or i['type'] == 'bool':
            if i['type'] == 'bool':
                if value not in (True, False):
                    raise ValueError('Config value %s must be True or False' % key)
            elif i['type'] in ('float', 'int'):
                value = (float if i['type'] == 'float' else int)(value)
                if 'min' in i and i['min'] is not None and value < i['min']:
                    raise ValueError('Config value %s must be at least %s' % (key, i['min']))
                if 'max' in i and i['max'] is not None and value > i['max']:
                    raise ValueError('Config value %s must not exceed %s' % (key, i['max']))
            elif i['type'] == 'enum':
                if value not in i['enum']:
                    import difflib
                    sim = difflib.get_close_matches(str(value), [str(e) for e in i['enum']], n=3, cutoff=.5)
                    raise ValueError('Wrong configuration, possible values are: %s%s' %
                                     (i['enum'], ('\nDid you mean: ' + ', '.join(sim)) if sim else ''))
        self._config[key]['value'] = value
