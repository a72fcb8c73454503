"""Coefficient (polynomial regression) application: mirror of the reference's ``coefficient`` package."""
