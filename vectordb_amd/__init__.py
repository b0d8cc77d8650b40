"""MI355X-native search core behind vectordb's distance and index interfaces."""

__version__ = "0.1.0"
