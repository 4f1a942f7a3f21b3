"""Test / bench scaffolding: synthetic scenes and mirrors of the reference's boundary helpers.  Not product code."""
