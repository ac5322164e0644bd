"""Reference-side bindings: what a MagicDrive / diffusers maintainer would add to call libmdx from the reference's own module tree."""
