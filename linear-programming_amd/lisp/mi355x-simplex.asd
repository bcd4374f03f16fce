;;;; mi355x-simplex.asd -- ASDF system of the MI355X dense-simplex backend.
;;;;
;;;; Loads next to the reference library (which stays untouched): the backend is just a new
;;;; value for linear-programming:*solver* (src/solver.lisp:39-56).
;;;;
;;;; NOTE: written against the reference's exported interface and the C ABI in
;;;; include/mi355x_simplex.h; it could not be executed in the build image (no Common Lisp
;;;; implementation is installed there).  tests/ drive the same C ABI call sequence from
;;;; Python/ctypes instead.
(asdf:defsystem "mi355x-simplex"
  :description "MI355X (gfx950) dense-simplex backend for linear-programming's *solver* hook"
  :version "0.1.0"
  :license "MIT"
  :depends-on ("linear-programming" "cffi")
  :components ((:file "mi355x-simplex")))
