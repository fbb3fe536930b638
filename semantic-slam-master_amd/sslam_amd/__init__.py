"""MI355X-native host layer of the semantic-slam extraction + matching hot path."""
from . import lib  # noqa: F401
