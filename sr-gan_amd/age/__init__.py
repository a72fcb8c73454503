"""Age estimation application: mirror of the reference's ``age`` package."""
