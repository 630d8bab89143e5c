"""MI355X-native Monte-Carlo ray-tracing core behind the Optics Design
Workbench API.  Importing the package never touches the GPU;
the HIP library is loaded on first use and its absence is an error, never a
silent CPU fallback."""
__version__ = '0.1.0'
