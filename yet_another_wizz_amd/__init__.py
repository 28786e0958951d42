"""yet_another_wizz_amd -- MI355X-native angular pair counting behind the yet_another_wizz API."""
