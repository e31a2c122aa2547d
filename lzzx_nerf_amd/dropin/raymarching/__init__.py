from .raymarching import *  # noqa: F401,F403  (the reference does `import raymarching; raymarching.march_rays(...)`)
