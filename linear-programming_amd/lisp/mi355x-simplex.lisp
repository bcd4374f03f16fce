;;;; mi355x-simplex.lisp -- CFFI glue: linear-programming:*solver*  ->  libmi355x_simplex.so
;;;;
;;;; Usage (nothing in the reference library changes):
;;;;
;;;;   (asdf:load-system "mi355x-simplex")
;;;;   (let ((linear-programming:*solver* 'mi355x-simplex:mi355x-simplex-solver))
;;;;     (linear-programming:solve-problem problem))          ; src/solver.lisp:53-56
;;;;
;;;; or (setf linear-programming:*solver* 'mi355x-simplex:mi355x-simplex-solver), after which
;;;; solve-problem / with-solved-problem / with-solution-variables work unchanged.
;;;;
;;;; Two routes from a `problem` to a solution object:
;;;;  * NATIVE (the default for problems whose numbers are double-floats / integers): the parsed
;;;;    problem (src/problem.lisp:45-53) is marshalled straight into the library (mi355x_problem_*),
;;;;    which assembles the tableau in C++ (= build-tableau, src/simplex.lisp:142-328, in double-float;
;;;;    single-phase problems never exist as a dense tableau on either side), solves it on the GPU and
;;;;    keeps a light solution (objective row, RHS column, basis, var-mapping: all that
;;;;    src/simplex.lisp:74-120 read).  The Lisp side holds a MI355X-SOLUTION with methods on the
;;;;    four solution-* generics (src/solver.lisp:59-80).  No boxed R x C matrix, no per-element
;;;;    coerce.
;;;;  * BUILD-TABLEAU route (:full-tableau t, :devices > 1, :native nil, or numbers the double-float
;;;;    assembly would not reproduce bit for bit -- ratios, single-floats): the reference's exported
;;;;    build-tableau, the tableau coerced and uploaded, the solved values written back into the
;;;;    same `tableau`, which the reference's own methods (src/solver.lisp:61-80) serve.
;;;; What stays in Lisp either way: the DSL parser, the hook, the generics.  What moves to the GPU:
;;;; n-solve-tableau (src/simplex.lisp:399-461) -- pricing, ratio test, rank-1 update and the
;;;; two-phase hand-over -- in double-float arithmetic.
;;;;
;;;; Scope: LP, double-float.  Problems with integer / binary variables are declined with
;;;; unsupported-constraint-error (src/conditions.lisp:69-77) as the hook's contract expects of
;;;; a backend (src/solver.lisp:40-45); rational problems are solved in double-float (every
;;;; entry is coerced), which is the documented behaviour of this backend.
;;;;
;;;; NOTE: could not be executed in the build image (no Lisp there); reviewed against
;;;; include/mi355x_simplex.h and the reference sources cited inline.

(defpackage :mi355x-simplex
  (:use :cl)
  (:import-from :linear-programming/simplex
                #:build-tableau #:tableau-matrix #:tableau-basis-columns
                #:tableau-var-count #:tableau-constraint-count #:tableau-instance-problem)
  (:import-from :linear-programming/problem
                #:problem-type #:problem-vars #:problem-objective-var #:problem-objective-func
                #:problem-integer-vars #:problem-var-bounds #:problem-constraints)
  (:import-from :linear-programming/solver
                #:solution-problem #:solution-objective-value #:solution-variable
                #:solution-reduced-cost)
  (:import-from :linear-programming/conditions
                #:solver-error #:unbounded-problem-error #:infeasible-problem-error
                #:unsupported-constraint-error)
  (:export #:mi355x-simplex-solver
           #:mi355x-solve-problems
           #:mi355x-solution
           #:solution-pivots
           #:free-solution
           #:native-var-mapping
           #:device-count
           #:mi355x-error))

(in-package :mi355x-simplex)

;;; ------------------------------------------------------------------ the shared library
(cffi:define-foreign-library libmi355x-simplex
  (:unix (:or "libmi355x_simplex.so" "./libmi355x_simplex.so"))
  (t (:default "libmi355x_simplex")))

(cffi:use-foreign-library libmi355x-simplex)

;; status codes, include/mi355x_simplex.h
(defconstant +mi-optimal+ 0)
(defconstant +mi-unbounded+ 1)
(defconstant +mi-infeasible+ 2)
(defconstant +mi-max-pivots+ 3)
(defconstant +mi-art-nonzero+ 4)
(defconstant +mi-art-stuck+ 5)
(defconstant +mi-nonfinite+ 6)   ; column shards only: the tableau overflowed (see solve-column-partitioned)
(defconstant +mi-cancelled+ 7)   ; mi355x_*_cancel from another thread (never from this single-threaded glue)
(defconstant +mi-running+ 100)   ; per-LP status of a batch member a capped / cancelled solve left unfinished

(cffi:defcfun ("mi355x_device_count" device-count) :int)
(cffi:defcfun ("mi355x_last_error" %last-error) :string)
(cffi:defcfun ("mi355x_tab_create" %tab-create) :int
  (out :pointer) (rows :int64) (cols :int64) (host-matrix :pointer) (host-basis :pointer)
  (device :int))
(cffi:defcfun ("mi355x_tab_create_compact" %tab-create-compact) :int
  (out :pointer) (rows :int64) (var-count :int64) (n-stored :int64) (host-stored :pointer)
  (stored-cols :pointer) (host-basis :pointer) (device :int))
(cffi:defcfun ("mi355x_tab_destroy" %tab-destroy) :void (tab :pointer))
(cffi:defcfun ("mi355x_tab_solve" %tab-solve) :int
  (tab :pointer) (is-max :int) (fp-factor :double) (max-pivots :int64) (n-pivots :pointer))
(cffi:defcfun ("mi355x_solve_two_phase" %solve-two-phase) :int
  (art :pointer) (main :pointer) (main-is-max :int) (fp-factor :double) (n-pivots :pointer))
(cffi:defcfun ("mi355x_two_phase_handover" %two-phase-handover) :int
  (art :pointer) (main :pointer) (fp-factor :double) (n-driveout :pointer))
(cffi:defcfun ("mi355x_tab_download" %tab-download) :int
  (tab :pointer) (host-matrix :pointer) (host-basis :pointer) (last-row :pointer)
  (last-col :pointer))
;; one tableau column-partitioned over several GPUs (include/mi355x_simplex.h, mi355x_colpart_*)
(cffi:defcfun ("mi355x_colpart_create_on" %colpart-create-on) :int
  (out :pointer) (rows :int64) (cols :int64) (host-matrix :pointer) (host-basis :pointer)
  (n-devices :int) (device-ids :pointer))
(cffi:defcfun ("mi355x_colpart_solve_two_phase" %colpart-solve-two-phase) :int
  (art :pointer) (main-cols :int64) (main-objective-row :pointer) (main-is-max :int)
  (fp-factor :double) (n-pivots :pointer) (main-out :pointer))
(cffi:defcfun ("mi355x_colpart_solve" %colpart-solve) :int
  (handle :pointer) (is-max :int) (fp-factor :double) (max-pivots :int64) (n-pivots :pointer))
(cffi:defcfun ("mi355x_colpart_download" %colpart-download) :int
  (handle :pointer) (host-matrix :pointer) (host-basis :pointer) (last-row :pointer)
  (last-col :pointer))
(cffi:defcfun ("mi355x_colpart_destroy" %colpart-destroy) :void (handle :pointer))
;; a batch of same-shape tableaux over one or several GPUs (include/mi355x_simplex.h, mi355x_multibatch_*)
(cffi:defcfun ("mi355x_multibatch_create" %multibatch-create) :int
  (out :pointer) (n-lps :int64) (rows :int64) (cols :int64) (host-matrices :pointer)
  (host-bases :pointer) (n-devices :int) (device-ids :pointer))
(cffi:defcfun ("mi355x_multibatch_solve" %multibatch-solve) :int
  (handle :pointer) (is-max :int) (fp-factor :double) (max-pivots :int64) (status :pointer)
  (n-pivots :pointer))
(cffi:defcfun ("mi355x_multibatch_solve_two_phase" %multibatch-solve-two-phase) :int
  (art :pointer) (main :pointer) (main-is-max :int) (fp-factor :double) (status :pointer)
  (n-pivots :pointer))
(cffi:defcfun ("mi355x_multibatch_two_phase_handover" %multibatch-two-phase-handover) :int
  (art :pointer) (main :pointer) (fp-factor :double) (phase1-status :pointer) (status :pointer)
  (n-driveout :pointer))
(cffi:defcfun ("mi355x_multibatch_download" %multibatch-download) :int
  (handle :pointer) (lp-index :int64) (host-matrix :pointer) (host-basis :pointer)
  (last-row :pointer) (last-col :pointer))
(cffi:defcfun ("mi355x_multibatch_destroy" %multibatch-destroy) :void (handle :pointer))

;; the native route: problem -> (C++ build-tableau, GPU solve) -> light solution
;; (include/mi355x_simplex.h, "native host side of the hook")
(cffi:defcfun ("mi355x_problem_create" %problem-create) :int
  (out :pointer) (is-max :int) (n-vars :int64))
(cffi:defcfun ("mi355x_problem_set_objective" %problem-set-objective) :int
  (problem :pointer) (vars :pointer) (coefs :pointer) (nnz :int64))
(cffi:defcfun ("mi355x_problem_set_bounds" %problem-set-bounds) :int
  (problem :pointer) (var :int64) (has-lb :int) (lb :double) (has-ub :int) (ub :double))
(cffi:defcfun ("mi355x_problem_set_integer" %problem-set-integer) :int
  (problem :pointer) (var :int64))
(cffi:defcfun ("mi355x_problem_add_constraint" %problem-add-constraint) :int
  (problem :pointer) (op :int) (vars :pointer) (coefs :pointer) (nnz :int64) (rhs :double))
(cffi:defcfun ("mi355x_problem_destroy" %problem-destroy) :void (problem :pointer))
(cffi:defcfun ("mi355x_var_mapping" %var-mapping) :int
  (problem :pointer) (var :int64) (kind :pointer) (col :pointer) (offset :pointer))
(cffi:defcfun ("mi355x_simplex_solver" %simplex-solver) :int
  (problem :pointer) (fp-tolerance :double) (device :int) (out :pointer))
(cffi:defcfun ("mi355x_simplex_solver_begin" %solver-begin) :int
  (problem :pointer) (fp-tolerance :double) (device :int) (out :pointer))
(cffi:defcfun ("mi355x_simplex_solver_step" %solver-step) :int
  (job :pointer) (max-pivots :int64) (n-pivots :pointer))
(cffi:defcfun ("mi355x_simplex_solver_finish" %solver-finish) :int
  (job :pointer) (out :pointer))
(cffi:defcfun ("mi355x_simplex_solver_abandon" %solver-abandon) :void (job :pointer))
(cffi:defcfun ("mi355x_simplex_solver_many_begin" %solver-many-begin) :int
  (problems :pointer) (n :int64) (fp-tolerance :double) (n-devices :int) (device-ids :pointer)
  (out :pointer))
(cffi:defcfun ("mi355x_simplex_solver_many_step" %solver-many-step) :int
  (job :pointer) (max-pivots :int64) (status :pointer))
(cffi:defcfun ("mi355x_simplex_solver_many_finish" %solver-many-finish) :int
  (job :pointer) (status :pointer) (out :pointer))
(cffi:defcfun ("mi355x_simplex_solver_many_abandon" %solver-many-abandon) :void (job :pointer))
(cffi:defcfun ("mi355x_solution_objective_value" %solution-objective-value) :int
  (solution :pointer) (out :pointer))
(cffi:defcfun ("mi355x_solution_variable" %solution-variable) :int
  (solution :pointer) (var :int64) (out :pointer))
(cffi:defcfun ("mi355x_solution_reduced_cost" %solution-reduced-cost) :int
  (solution :pointer) (var :int64) (out :pointer))
(cffi:defcfun ("mi355x_solution_pivots" %solution-pivots) :int
  (solution :pointer) (phase1 :pointer) (phase2 :pointer))
(cffi:defcfun ("mi355x_solution_destroy" %solution-destroy) :void (solution :pointer))

(define-condition mi355x-error (solver-error)
  ((code :initarg :code :reader mi355x-error-code)
   (message :initarg :message :reader mi355x-error-message))
  (:report (lambda (err stream)
             (format stream "libmi355x_simplex failed with status ~D: ~A"
                     (mi355x-error-code err) (mi355x-error-message err)))))

(defun check (code)
  "Negative statuses are library errors (MI_BAD_ARG, MI_HIP_ERROR, MI_NO_DEVICE ...)."
  (when (minusp code)
    (error 'mi355x-error :code code :message (%last-error)))
  code)

(defmacro with-foreign-fp-mode (&body body)
  "SBCL enables floating point traps; foreign code that produces an inf/nan must not raise
SIGFPE in the Lisp thread."
  #+sbcl `(sb-int:with-float-traps-masked (:overflow :invalid :divide-by-zero :inexact)
            ,@body)
  #-sbcl `(progn ,@body))

;;; ------------------------------------------------------------------ tableau <-> raw doubles
(defun tableau->vectors (tableau)
  "Flatten the boxed (simple-array real 2) of src/simplex.lisp:53 into a row-major
(simple-array double-float (*)) and the fixnum basis (tagged words: never hand their storage
to C) into a (signed-byte 64) vector."
  (let* ((matrix (tableau-matrix tableau))
         (rows (array-dimension matrix 0))
         (cols (array-dimension matrix 1))
         (flat (make-array (* rows cols) :element-type 'double-float))
         (basis-src (tableau-basis-columns tableau))
         (basis (make-array (max 1 (length basis-src)) :element-type '(signed-byte 64)
                                                       :initial-element 0)))
    (dotimes (r rows)
      (dotimes (c cols)
        (setf (aref flat (+ (* r cols) c))
              (coerce (aref matrix r c) 'double-float))))
    (dotimes (i (length basis-src))
      (setf (aref basis i) (aref basis-src i)))
    (values flat basis rows cols)))

(defun vectors->tableau (tableau flat basis)
  "Write the solved values back into the tableau's own arrays (the struct slots are read-only,
the arrays are not), so tableau-variable & co. (src/simplex.lisp:74-120) read GPU results."
  (let* ((matrix (tableau-matrix tableau))
         (rows (array-dimension matrix 0))
         (cols (array-dimension matrix 1))
         (basis-dst (tableau-basis-columns tableau)))
    (dotimes (r rows)
      (dotimes (c cols)
        (setf (aref matrix r c) (aref flat (+ (* r cols) c)))))
    (dotimes (i (length basis-dst))
      (setf (aref basis-dst i) (aref basis i)))
    tableau))

(defun upload-tableau (tableau device)
  "Returns (values handle flat basis): a device handle plus the host staging vectors."
  (multiple-value-bind (flat basis rows cols) (tableau->vectors tableau)
    (cffi:with-foreign-object (out :pointer)
      (cffi:with-pointer-to-vector-data (pm flat)
        (cffi:with-pointer-to-vector-data (pb basis)
          (check (with-foreign-fp-mode
                   (%tab-create out rows cols pm pb device)))))
      (values (cffi:mem-ref out :pointer) flat basis))))

(defun exact-unit-entry-p (x one)
  "X is exactly 1 (ONE true) / exactly +0 (ONE false): the integers build-tableau stores
(src/simplex.lisp:217-219, 255-259) or their float forms, never -0.0."
  (if one
      (= x 1)
      (and (zerop x) (not (and (floatp x) (minusp (float-sign x)))))))

(defun unit-basis-p (tableau)
  "True when every basis column is exactly the unit vector of its row (objective row included):
what build-tableau produces for the slack columns of a single-phase problem."
  (let* ((matrix (tableau-matrix tableau))
         (rows (array-dimension matrix 0))
         (var-count (1- (array-dimension matrix 1)))
         (basis (tableau-basis-columns tableau))
         (seen (make-array var-count :element-type 'bit :initial-element 0)))
    (dotimes (i (length basis) t)
      (let ((b (aref basis i)))
        (unless (and (integerp b) (<= 0 b) (< b var-count) (zerop (aref seen b)))
          (return nil))
        (setf (aref seen b) 1)
        (dotimes (r rows)
          (unless (exact-unit-entry-p (aref matrix r b) (= r i))
            (return-from unit-basis-p nil)))))))

(defun upload-tableau-compact (tableau device)
  "Single-phase problems: only [non-basic columns | RHS] is coerced and crosses PCIe
(mi355x_tab_create_compact); the slack identity block -- a third of the boxed matrix at n = 2m --
is neither converted nor uploaded, and no dense device buffer exists for a plain solve with the
light read-back.  Returns (values handle basis), or NIL when the basis columns are not exact unit
vectors (the caller then uploads the dense tableau)."
  (when (and (plusp (length (tableau-basis-columns tableau))) (unit-basis-p tableau))
    (let* ((matrix (tableau-matrix tableau))
           (rows (array-dimension matrix 0))
           (var-count (1- (array-dimension matrix 1)))
           (basis-src (tableau-basis-columns tableau))
           (m (length basis-src))
           (n-stored (- var-count m))
           (basic (make-array var-count :element-type 'bit :initial-element 0)))
      (when (plusp n-stored)
        (dotimes (i m) (setf (aref basic (aref basis-src i)) 1))
        (let ((stored-cols (make-array n-stored :element-type '(signed-byte 64)))
              (stored (make-array (* rows (1+ n-stored)) :element-type 'double-float))
              (basis (make-array m :element-type '(signed-byte 64)))
              (j 0))
          (dotimes (c var-count)
            (when (zerop (aref basic c))
              (setf (aref stored-cols j) c)
              (incf j)))
          (dotimes (r rows)
            (let ((base (* r (1+ n-stored))))
              (dotimes (k n-stored)
                (setf (aref stored (+ base k))
                      (coerce (aref matrix r (aref stored-cols k)) 'double-float)))
              (setf (aref stored (+ base n-stored))
                    (coerce (aref matrix r var-count) 'double-float))))
          (dotimes (i m) (setf (aref basis i) (aref basis-src i)))
          (cffi:with-foreign-object (out :pointer)
            (cffi:with-pointer-to-vector-data (ps stored)
              (cffi:with-pointer-to-vector-data (pc stored-cols)
                (cffi:with-pointer-to-vector-data (pb basis)
                  (check (with-foreign-fp-mode
                           (%tab-create-compact out rows var-count n-stored ps pc pb device))))))
            (values (cffi:mem-ref out :pointer) basis)))))))

(defun download-tableau (handle tableau flat basis)
  "Full write-back: every entry of the solved tableau (400 MB and 5e7 boxed doubles at
8192 x 4096 -- only worth it when the caller wants to look inside the tableau)."
  (cffi:with-pointer-to-vector-data (pm flat)
    (cffi:with-pointer-to-vector-data (pb basis)
      (check (%tab-download handle pm pb (cffi:null-pointer) (cffi:null-pointer)))))
  (vectors->tableau tableau flat basis))

(defun download-solution (handle tableau basis)
  "Light write-back (the default): the objective row, the RHS column and the basis are all
that tableau-objective-value, tableau-variable and tableau-reduced-cost read
(src/simplex.lisp:74-120), so only those are fetched and stored into the tableau's arrays;
the interior of the matrix keeps its pre-solve contents."
  (let* ((matrix (tableau-matrix tableau))
         (rows (array-dimension matrix 0))
         (cols (array-dimension matrix 1))
         (last-row (make-array cols :element-type 'double-float))
         (last-col (make-array rows :element-type 'double-float))
         (basis-dst (tableau-basis-columns tableau)))
    (cffi:with-pointer-to-vector-data (pr last-row)
      (cffi:with-pointer-to-vector-data (pc last-col)
        (cffi:with-pointer-to-vector-data (pb basis)
          (check (%tab-download handle (cffi:null-pointer) pb pr pc)))))
    (dotimes (r rows)
      (setf (aref matrix r (1- cols)) (aref last-col r)))
    (dotimes (c cols)
      (setf (aref matrix (1- rows) c) (aref last-row c)))
    (dotimes (i (length basis-dst))
      (setf (aref basis-dst i) (aref basis i)))
    tableau))

;;; ------------------------------------------------------------------ several GPUs
(defun device-count-of (devices)
  "DEVICES is a count (devices 0 .. n-1) or a list of device ids."
  (if (listp devices) (length devices) devices))

(defun colpart-create (flat basis rows cols devices)
  "mi355x_colpart_create_on: one shard per device of DEVICES (a count, or a list of distinct ids)."
  (let ((n (device-count-of devices)))
    (cffi:with-foreign-object (out :pointer)
      (cffi:with-foreign-object (ids :int (max n 1))
        (when (listp devices)
          (loop for d in devices for i from 0 do (setf (cffi:mem-aref ids :int i) d)))
        (cffi:with-pointer-to-vector-data (pm flat)
          (cffi:with-pointer-to-vector-data (pb basis)
            (check (with-foreign-fp-mode
                     (%colpart-create-on out rows cols pm pb n
                                         (if (listp devices) ids (cffi:null-pointer))))))))
      (cffi:mem-ref out :pointer))))

(defun colpart-read-back (handle tableau flat basis rows cols full-tableau)
  "Full or light write-back from a column-partitioned handle into TABLEAU's arrays."
  (let* ((matrix (tableau-matrix tableau))
         (last-row (make-array cols :element-type 'double-float))
         (last-col (make-array rows :element-type 'double-float))
         (basis-dst (tableau-basis-columns tableau)))
    (cffi:with-pointer-to-vector-data (pr last-row)
      (cffi:with-pointer-to-vector-data (pc last-col)
        (cffi:with-pointer-to-vector-data (pb basis)
          (if full-tableau
              (cffi:with-pointer-to-vector-data (pm flat)
                (check (%colpart-download handle pm pb pr pc)))
              (check (%colpart-download handle (cffi:null-pointer) pb pr pc))))))
    (if full-tableau
        (vectors->tableau tableau flat basis)
        (progn
          (dotimes (r rows) (setf (aref matrix r (1- cols)) (aref last-col r)))
          (dotimes (c cols) (setf (aref matrix (1- rows) c) (aref last-row c)))
          (dotimes (i (length basis-dst)) (setf (aref basis-dst i) (aref basis i)))
          tableau))))

(defun solve-two-phase-column-partitioned (art-tab main-tab devices factor full-tableau n-pivots)
  "Two-phase n-solve-tableau (src/simplex.lisp:402-452) with the ARTIFICIAL tableau
column-partitioned over DEVICES: phase 1, the drive-out pivots, the hand-over and phase 2 all stay
on the partition (mi355x_colpart_solve_two_phase).  The main tableau is not uploaded -- the library
takes its objective row only; its constraint rows are the artificial tableau's.  Returns
:unsupported / :overflowed when the partitioned path does not apply (the caller's tableaux are
untouched and it solves them on one device)."
  (multiple-value-bind (art-flat art-basis rows art-cols) (tableau->vectors art-tab)
    (let* ((main-matrix (tableau-matrix main-tab))
           (main-cols (array-dimension main-matrix 1))
           (objective (make-array main-cols :element-type 'double-float))
           (art-handle (colpart-create art-flat art-basis rows art-cols devices))
           (main-handle (cffi:null-pointer)))
      (dotimes (c main-cols)
        (setf (aref objective c) (coerce (aref main-matrix (1- rows) c) 'double-float)))
      (unwind-protect
           (let ((status
                   (cffi:with-foreign-object (out :pointer)
                     (setf (cffi:mem-ref out :pointer) (cffi:null-pointer))
                     (prog1
                         (cffi:with-pointer-to-vector-data (po objective)
                           (with-foreign-fp-mode
                             (%colpart-solve-two-phase art-handle main-cols po
                                                       (max-problem-p main-tab) factor n-pivots
                                                       out)))
                       (setf main-handle (cffi:mem-ref out :pointer))))))
             (cond
               ((= status -6) :unsupported)                 ; MI_UNSUPPORTED: artificial basis not unit columns
               ((= status +mi-nonfinite+) :overflowed)
               (t
                (check status)
                (signal-outcome status)
                (let ((main-flat (make-array (* rows main-cols) :element-type 'double-float))
                      (main-basis (make-array (max 1 (1- rows)) :element-type '(signed-byte 64)
                                                                :initial-element 0)))
                  (colpart-read-back main-handle main-tab main-flat main-basis rows main-cols
                                     full-tableau)))))
        (unless (cffi:null-pointer-p main-handle) (%colpart-destroy main-handle))
        (%colpart-destroy art-handle)))))

(defun solve-column-partitioned (tableau devices factor max-pivots full-tableau n-pivots)
  "Single-phase n-solve-tableau (src/simplex.lisp:453-461) with the tableau's non-basic columns
distributed over DEVICES GPUs of this node (RCCL over xGMI inside the library; logical shards on
one GPU when fewer are visible).  Same pivots, same bits as on one device."
  (multiple-value-bind (flat basis rows cols) (tableau->vectors tableau)
    (let ((handle (colpart-create flat basis rows cols devices)))
      (unwind-protect
           (let ((status (solve-in-chunks
                          (lambda (cap) (%colpart-solve handle (max-problem-p tableau) factor cap n-pivots))
                          rows cols max-pivots n-pivots)))
             ;; A compact column shard cannot do what the reference does with an entering column
             ;; that holds an infinity or a NaN (it would turn basic columns, which a shard does not
             ;; store, into NaNs): the library stops with MI_NONFINITE and the caller, whose tableau
             ;; has not been written to, solves it on one device, where that case is reproduced.
             (when (= status +mi-nonfinite+)
               (return-from solve-column-partitioned :overflowed))
             (signal-outcome status)
             (colpart-read-back handle tableau flat basis rows cols full-tableau))
        (%colpart-destroy handle)))))

(defun signal-outcome (status)
  "C outcome -> the reference's conditions (src/conditions.lisp:43-60)."
  (cond
    ((= status +mi-optimal+) nil)
    ((= status +mi-unbounded+) (error 'unbounded-problem-error))     ; src/simplex.lisp:458-459
    ((= status +mi-infeasible+) (error 'infeasible-problem-error))   ; src/simplex.lisp:405-407
    ((= status +mi-art-nonzero+) (error "Artificial variable still non-zero"))
    ((= status +mi-art-stuck+)
     (error "Artificial variable still in basis and cannot be replaced"))
    ((= status +mi-max-pivots+) (error 'mi355x-error :code status :message "pivot cap reached"))
    (t (error 'mi355x-error :code status :message "unknown status"))))

(defun max-problem-p (tableau)
  (if (eq 'max (problem-type (tableau-instance-problem tableau))) 1 0))

;;; ------------------------------------------------------------------ a way out of a solve
;;; The reference's loop has no iteration cap and no anti-cycling rule (src/simplex.lisp:453-461):
;;; an LP that cycles under Dantzig's rule with lowest-index ties runs for ever.  In Lisp that
;;; loop can be interrupted (C-c, sb-ext:with-timeout, bt:interrupt-thread); one blocking foreign
;;; call cannot.  So the glue never makes an unbounded foreign call for a single-phase solve: it
;;; asks for CHUNK pivots at a time (MI_MAX_PIVOTS = "chunk used up, still running") and is back
;;; in Lisp -- where pending interrupts are served -- a few times per second.  A solve continued
;;; call by call takes exactly the pivots of one long call (the cap is a test in the select step;
;;; pinned by the test-suite through the Python mirror's identical loop).  Other threads can
;;; also end a blocking call with mi355x_tab_cancel / mi355x_colpart_cancel (MI_CANCELLED).
(defun chunk-pivots (rows cols)
  "Pivots per foreign call: about a tenth to half a second of GPU time at every size (one pivot
moves 16 rows x cols bytes through HBM per 16 pivots; small tableaux cost ~4 us per pivot)."
  (max 1024 (min 65536 (floor (expt 2 36) (max 1 (* rows cols))))))

(defun solve-in-chunks (solve-fn rows cols max-pivots n-pivots)
  "Calls (funcall SOLVE-FN cap) -- cap pivots of n-solve-tableau, *N-PIVOTS = pivots of that call
-- until the status is something else than MI_MAX_PIVOTS or MAX-PIVOTS (0 = no cap) are used up.
Leaves the total in N-PIVOTS[0] and returns the last status."
  (let ((chunk (chunk-pivots rows cols))
        (total 0))
    (loop
      (let* ((cap (if (plusp max-pivots) (min chunk (- max-pivots total)) chunk))
             (status (check (with-foreign-fp-mode (funcall solve-fn cap)))))
        (incf total (cffi:mem-aref n-pivots :int64 0))
        (when (or (/= status +mi-max-pivots+)
                  (and (plusp max-pivots) (>= total max-pivots)))
          (setf (cffi:mem-aref n-pivots :int64 0) total)
          (return status))))))

;;; ------------------------------------------------------------------ the native route
;;; SURVEY 8(f) rows 2-3: at 8192 x 4096 the reference's build-tableau conses a boxed 4097 x 12289
;;; (simple-array real 2) -- 5e7 boxed entries -- and the build-tableau route then coerces every one
;;; of them.  Here the parsed problem itself crosses the C ABI: one foreign call per constraint with
;;; its coefficients in two specialised vectors, the tableau is assembled by the library's host
;;; threads straight into pinned staging buffers (csrc/host_problem.cpp), and what comes back is a
;;; handle to (objective row, RHS column, basis, var-mapping).
(defclass mi355x-solution ()
  ((problem :initarg :problem :reader mi355x-solution-problem
            :documentation "The problem instance the solution answers (solution-problem).")
   (handle :initarg :handle :accessor mi355x-solution-handle
           :documentation "mi355x_solution*; a null pointer once FREE-SOLUTION has run.")
   (var-index :initarg :var-index :reader mi355x-solution-var-index
              :documentation "eq hash table: variable symbol -> its index in problem-vars, the
name the C ABI knows the variable by."))
  (:documentation "What MI355X-SIMPLEX-SOLVER returns on the native route: the light solution
object of the library behind the four solution-* generics (src/solver.lisp:59-80).  Holds
O(rows + cols) doubles on the C side; released by the garbage collector's finalizer (SBCL) or
explicitly by FREE-SOLUTION."))

(defun make-solution (problem handle var-index)
  (let ((solution (make-instance 'mi355x-solution :problem problem :handle handle
                                                  :var-index var-index)))
    ;; the finalizer must not close over SOLUTION itself (it would never become garbage)
    #+sbcl (sb-ext:finalize solution (lambda () (%solution-destroy handle)) :dont-save t)
    solution))

(defun free-solution (solution)
  "Releases the C side of SOLUTION now instead of at some later garbage collection.  Idempotent;
the solution answers no generic afterwards."
  (let ((handle (mi355x-solution-handle solution)))
    (unless (cffi:null-pointer-p handle)
      #+sbcl (sb-ext:cancel-finalization solution)
      (setf (mi355x-solution-handle solution) (cffi:null-pointer))
      (%solution-destroy handle)))
  nil)

(defun live-handle (solution)
  (let ((handle (mi355x-solution-handle solution)))
    (when (cffi:null-pointer-p handle)
      (error "~S has been released with free-solution" solution))
    handle))

(defun solution-var-index (solution variable)
  "Index of VARIABLE in problem-vars, or the reference's error for a stranger
(src/simplex.lisp:85-86, 115-116)."
  (or (gethash variable (mi355x-solution-var-index solution))
      (error "~S is not a variable in the tableau" variable)))

(defmethod solution-problem ((solution mi355x-solution))          ; src/solver.lisp:59-62
  (mi355x-solution-problem solution))

(defmethod solution-objective-value ((solution mi355x-solution))  ; src/solver.lisp:64-67, simplex.lisp:74-78
  (cffi:with-foreign-object (out :double)
    (check (%solution-objective-value (live-handle solution) out))
    (cffi:mem-ref out :double)))

(defmethod solution-variable ((solution mi355x-solution) variable) ; src/solver.lisp:69-72, simplex.lisp:81-107
  (if (eq variable (problem-objective-var (mi355x-solution-problem solution)))
      (solution-objective-value solution)
      (let ((index (solution-var-index solution variable)))
        (cffi:with-foreign-object (out :double)
          (check (%solution-variable (live-handle solution) index out))
          (cffi:mem-ref out :double)))))

(defmethod solution-reduced-cost ((solution mi355x-solution) variable) ; src/solver.lisp:74-80, simplex.lisp:111-120
  (let ((index (solution-var-index solution variable)))
    (cffi:with-foreign-object (out :double)
      (let ((status (%solution-reduced-cost (live-handle solution) index out)))
        ;; MI_BAD_ARG with a valid index: the variable's mapping is not `positive`
        (when (= status -1)
          (error "~S has no lower bound" variable))                 ; src/simplex.lisp:117-118
        (check status)
        (cffi:mem-ref out :double)))))

(defun solution-pivots (solution)
  "(values phase-1-pivots phase-2-pivots) the solve took (drive-out pivots count as phase 1)."
  (cffi:with-foreign-objects ((p1 :int64) (p2 :int64))
    (check (%solution-pivots (live-handle solution) p1 p2))
    (values (cffi:mem-ref p1 :int64) (cffi:mem-ref p2 :int64))))

(defun call-with-linear-expression (alist var-index fn)
  "ALIST: a linear expression ((var . coef) ...) as the parser leaves it (src/problem.lisp:45-53).
Calls FN with pointers to its variable indices (int64) and coefficients (double) and their count;
the vectors are pinned for the length of the call only (the library copies)."
  (let* ((nnz (length alist))
         (vars (make-array (max nnz 1) :element-type '(signed-byte 64) :initial-element 0))
         (coefs (make-array (max nnz 1) :element-type 'double-float :initial-element 0d0)))
    (loop for (var . coef) in alist for k from 0
          do (setf (aref vars k) (or (gethash var var-index)
                                     (error "~S is not a variable of the problem" var))
                   (aref coefs k) (coerce coef 'double-float)))
    (cffi:with-pointer-to-vector-data (pv vars)
      (cffi:with-pointer-to-vector-data (pc coefs)
        (funcall fn pv pc nnz)))))

(defun marshal-problem (problem)
  "PROBLEM (src/problem.lisp:45-53) -> (values mi355x_problem* var-index): variables named by
their index in problem-vars, every number coerced to double-float.  The caller destroys the
handle (mi355x_problem_destroy)."
  (let* ((vars (problem-vars problem))
         (var-index (make-hash-table :test #'eq :size (max 1 (length vars)))))
    (loop for var across vars for i from 0 do (setf (gethash var var-index) i))
    (cffi:with-foreign-object (out :pointer)
      (check (%problem-create out (if (eq 'max (problem-type problem)) 1 0) (length vars)))
      (let ((handle (cffi:mem-ref out :pointer))
            (complete nil))
        (unwind-protect
             (progn
               (call-with-linear-expression
                (problem-objective-func problem) var-index
                (lambda (pv pc nnz) (check (%problem-set-objective handle pv pc nnz))))
               ;; (var . (lb . ub)), NIL = unbounded on that side; a variable without an entry
               ;; is >= 0 (src/simplex.lisp:189-212)
               (loop for (var . (lb . ub)) in (problem-var-bounds problem)
                     do (check (%problem-set-bounds
                                handle (or (gethash var var-index)
                                           (error "~S is not a variable of the problem" var))
                                (if lb 1 0) (if lb (coerce lb 'double-float) 0d0)
                                (if ub 1 0) (if ub (coerce ub 'double-float) 0d0))))
               (dolist (var (problem-integer-vars problem))
                 (check (%problem-set-integer handle (gethash var var-index))))
               ;; (op alist rhs), op one of <= >= = (src/problem.lisp:45-53)
               (loop for (op expression rhs) in (problem-constraints problem)
                     do (call-with-linear-expression
                         expression var-index
                         (lambda (pv pc nnz)
                           (check (%problem-add-constraint handle (ecase op (<= 0) (>= 1) (= 2))
                                                           pv pc nnz (coerce rhs 'double-float))))))
               (setf complete t)
               (values handle var-index))
          (unless complete (%problem-destroy handle)))))))

(defun native-var-mapping (problem variable)
  "The var-mapping entry the library's build-tableau gives VARIABLE, in the reference's form
(src/simplex.lisp:44-46): (positive col offset), (negative col offset) or (signed col) -- with the
reference's OWN symbols (internal to linear-programming/simplex, src/simplex.lisp:197-210), so that an
entry compares EQUAL with what (gethash variable (tableau-var-mapping tableau)) holds there."
  (multiple-value-bind (handle var-index) (marshal-problem problem)
    (unwind-protect
         (cffi:with-foreign-objects ((kind :int) (col :int64) (offset :double))
           (check (%var-mapping handle (or (gethash variable var-index)
                                           (error "~S is not a variable of the problem" variable))
                                kind col offset))
           (ecase (cffi:mem-ref kind :int)
             (0 (list 'linear-programming/simplex::positive (cffi:mem-ref col :int64) (cffi:mem-ref offset :double)))
             (1 (list 'linear-programming/simplex::negative (cffi:mem-ref col :int64) (cffi:mem-ref offset :double)))
             (2 (list 'linear-programming/simplex::signed (cffi:mem-ref col :int64)))))
      (%problem-destroy handle))))

(defun native-number-p (x)
  "Numbers the library's double-float build-tableau treats exactly as the reference's generic
arithmetic followed by the glue's coerce would: double-floats, and integers a double holds
exactly.  (A ratio or a single-float is combined in its own type first by the reference --
rhs - coef * lb at src/simplex.lisp:235-238 -- and rounded to double afterwards.)"
  (or (typep x 'double-float)
      (and (integerp x) (<= (abs x) (expt 2 53)))))

(defun native-numbers-p (problem)
  (flet ((expression-ok (alist) (every (lambda (term) (native-number-p (cdr term))) alist)))
    (and (expression-ok (problem-objective-func problem))
         (every (lambda (entry)
                  (destructuring-bind (lb . ub) (cdr entry)
                    (and (or (null lb) (native-number-p lb)) (or (null ub) (native-number-p ub)))))
                (problem-var-bounds problem))
         (every (lambda (constraint)
                  (and (expression-ok (second constraint)) (native-number-p (third constraint))))
                (problem-constraints problem)))))

(defun solve-natively (problem factor device max-pivots)
  "simplex-solver for an LP (src/simplex.lisp:506-542 without branch-and-bound) entirely behind
the C ABI: marshal, mi355x_simplex_solver_begin (build-tableau + upload), ..._step in bounded
chunks (never an unbounded foreign call; the phases of a two-phase problem included), ..._finish.
Returns a MI355X-SOLUTION or signals the reference's conditions."
  (multiple-value-bind (problem-handle var-index) (marshal-problem problem)
    (unwind-protect
         (cffi:with-foreign-objects ((out :pointer) (n-pivots :int64))
           (let ((status (with-foreign-fp-mode (%solver-begin problem-handle factor device out))))
             ;; the no-constraint special case decides unboundedness while building
             ;; (src/simplex.lisp:170, 174)
             (when (= status +mi-unbounded+) (error 'unbounded-problem-error))
             (check status))
           (let ((job (cffi:mem-ref out :pointer))
                 (consumed nil)
                 (rows (1+ (length (problem-constraints problem)))))
             (unwind-protect
                  (let ((status (solve-in-chunks
                                 (lambda (cap) (%solver-step job cap n-pivots))
                                 rows (+ rows (length (problem-vars problem))) max-pivots n-pivots)))
                    (signal-outcome status)
                    (setf consumed t)                ; finish consumes the job whatever it returns
                    (check (%solver-finish job out))
                    (make-solution problem (cffi:mem-ref out :pointer) var-index))
               (unless consumed (%solver-abandon job)))))
      (%problem-destroy problem-handle))))

(defun solve-problems-natively (problems factor devices max-pivots)
  "A LIST of problems entirely behind the C ABI: every problem marshalled (mi355x_problem_*), ONE job of
the library for the list (mi355x_simplex_solver_many_begin groups the members by tableau shape and
sense into multi-device batches; two-phase members as pairs of batches with the step between the
phases on the devices), stepped in bounded foreign calls, the light solutions read back.  Returns a
list parallel to PROBLEMS: a MI355X-SOLUTION, or the condition object the one-problem hook would
signal for that member.  No boxed tableau exists for any member (BASELINE config 4 from Lisp: 1 024
build-tableau results would be 2e8 boxed entries)."
  (let* ((n (length problems))
         (marshalled '())                ; (handle var-index) per problem, filled INSIDE the unwind-protect: a
         (n-dev (device-count-of devices))   ; marshalling error midway must not leak the handles made so far
         (rows (1+ (reduce #'max problems :key (lambda (p) (length (problem-constraints p))))))
         (cols (+ rows (reduce #'max problems :key (lambda (p) (length (problem-vars p))))))
         (chunk (chunk-pivots rows cols)))
    (unwind-protect
         (cffi:with-foreign-objects ((handles :pointer n) (ids :int (max n-dev 1)) (out :pointer)
                                     (status :int32 n) (solutions :pointer n))
           (dolist (problem problems)
             (push (multiple-value-list (marshal-problem problem)) marshalled))
           (setf marshalled (nreverse marshalled))
           (loop for (handle nil) in marshalled for k from 0
                 do (setf (cffi:mem-aref handles :pointer k) handle))
           (when (listp devices)
             (loop for d in devices for i from 0 do (setf (cffi:mem-aref ids :int i) d)))
           (check (with-foreign-fp-mode
                    (%solver-many-begin handles n factor n-dev
                                        (if (listp devices) ids (cffi:null-pointer)) out)))
           (let ((job (cffi:mem-ref out :pointer))
                 (consumed nil)
                 (done 0))
             (unwind-protect
                  (progn
                    (loop
                      (let* ((cap (if (plusp max-pivots) (min chunk (- max-pivots done)) chunk))
                             (rc (check (with-foreign-fp-mode (%solver-many-step job cap status)))))
                        (incf done cap)
                        (when (or (/= rc +mi-max-pivots+)
                                  (and (plusp max-pivots) (>= done max-pivots)))
                          (return))))
                    (setf consumed t)                ; finish consumes the job whatever it returns
                    (check (%solver-many-finish job status solutions))
                    (loop for problem in problems
                          for (nil var-index) in marshalled
                          for k from 0
                          collect (let ((st (cffi:mem-aref status :int32 k)))
                                    (cond
                                      ((= st +mi-optimal+)
                                       (make-solution problem (cffi:mem-aref solutions :pointer k) var-index))
                                      ((= st -6)                         ; MI_UNSUPPORTED: integer variables
                                       (make-condition 'unsupported-constraint-error
                                                       :constraint (cons 'integer (problem-integer-vars problem))
                                                       :solver-name "mi355x-simplex"))
                                      ((= st +mi-running+) (outcome-condition +mi-max-pivots+))
                                      ;; a member whose unit FAILED (device error, out of memory ...: the
                                      ;; other units went on) carries that negative code
                                      ((minusp st)
                                       (make-condition 'mi355x-error :code st
                                                                     :message (format nil "member ~D failed" k)))
                                      (t (outcome-condition st))))))
               (unless consumed (%solver-many-abandon job)))))
      (loop for (handle nil) in marshalled do (%problem-destroy handle)))))

;;; ------------------------------------------------------------------ the *solver* value
(defun solve-two-phase-in-chunks (art-handle main-handle rows cols main-is-max factor max-pivots
                                  n-pivots)
  "n-solve-tableau's two-phase branch (src/simplex.lisp:402-452) without an unbounded foreign call:
phase 1 in chunks on the artificial tableau, the step between the phases
(mi355x_two_phase_handover: feasibility test, drive-out pivots, hand-over -- bounded by the row
count), phase 2 in chunks on the main tableau.  Same pivots, same bits as mi355x_solve_two_phase.
ROWS x COLS: the artificial tableau's shape (chunk size).  MAX-PIVOTS (0 = none) caps the two
phases together.  Returns the final status."
  (let* ((used 0)
         (status (solve-in-chunks (lambda (cap) (%tab-solve art-handle 0 factor cap n-pivots))
                                  rows cols max-pivots n-pivots)))
    (incf used (cffi:mem-aref n-pivots :int64 0))
    (cond
      ((/= status +mi-optimal+) status)
      (t
       (setf status (check (with-foreign-fp-mode
                             (%two-phase-handover art-handle main-handle factor n-pivots))))
       (incf used (cffi:mem-aref n-pivots :int64 0))
       (cond
         ((/= status +mi-optimal+) status)         ; MI_INFEASIBLE / MI_ART_NONZERO / MI_ART_STUCK
         ((and (plusp max-pivots) (>= used max-pivots)) +mi-max-pivots+)
         (t (solve-in-chunks (lambda (cap) (%tab-solve main-handle main-is-max factor cap n-pivots))
                             rows cols (if (plusp max-pivots) (- max-pivots used) 0)
                             n-pivots)))))))

(defun mi355x-simplex-solver (problem &rest args
                              &key (fp-tolerance 1024) (device 0) (devices 1) (max-pivots 0)
                                full-tableau (native :auto)
                              &allow-other-keys)
  "Solver interface function for the MI355X backend (the value of
linear-programming:*solver*, src/solver.lisp:39-49).  Takes a problem and backend keyword
arguments -- :fp-tolerance (as the built-in solver, src/simplex.lisp:506-511), :device,
:devices (a count > 1 or a list of device ids: the tableau -- for a two-phase problem the
artificial tableau, with phase 1, the hand-over and phase 2 all on the partition -- is
column-partitioned over those GPUs of the node), :max-pivots, :full-tableau (return a `tableau`
with every entry of the solved tableau written back), :native (:auto, the default: the native
route whenever the problem's numbers are double-floats / integers and neither :full-tableau nor
:devices asks for the tableau itself; T: the native route also for ratios and single-floats, which
are then rounded to double BEFORE build-tableau's arithmetic instead of after; NIL: always the
build-tableau route) -- and returns a solution object answering solution-problem,
solution-objective-value, solution-variable and solution-reduced-cost (src/solver.lisp:40-45): a
MI355X-SOLUTION on the native route, a solved `tableau` otherwise.  solve-problem forwards these
keywords (src/solver.lisp:53-56):
  (solve-problem problem :devices 8)        (solve-problem problem :devices '(4 5 6 7))"
  (declare (ignore args))
  (when (problem-integer-vars problem)
    (error 'unsupported-constraint-error
           :constraint (cons 'integer (problem-integer-vars problem))
           :solver-name "mi355x-simplex"))
  (when (and native (not full-tableau) (<= (device-count-of devices) 1)
             (plusp (length (problem-vars problem)))
             (or (eq native t) (native-numbers-p problem)))
    (return-from mi355x-simplex-solver
      (solve-natively problem (coerce fp-tolerance 'double-float) device max-pivots)))
  (let ((tableaus (build-tableau problem problem :fp-tolerance-factor fp-tolerance))
        (factor (coerce fp-tolerance 'double-float)))
    (cffi:with-foreign-object (n-pivots :int64 2)
      (if (listp tableaus)
          ;; two-phase: (art-tableau main-tableau), src/simplex.lisp:326-328, 402-452
          (destructuring-bind (art-tab main-tab) tableaus
           (if (and (> (device-count-of devices) 1)
                    (not (member (solve-two-phase-column-partitioned art-tab main-tab devices factor
                                                                     full-tableau n-pivots)
                                 '(:unsupported :overflowed))))
               main-tab
            (multiple-value-bind (art-handle art-flat art-basis) (upload-tableau art-tab device)
              (declare (ignorable art-flat art-basis))
              (unwind-protect
                   (multiple-value-bind (main-handle main-flat main-basis)
                       (upload-tableau main-tab device)
                     (unwind-protect
                          (let ((status (solve-two-phase-in-chunks
                                         art-handle main-handle
                                         (array-dimension (tableau-matrix art-tab) 0)
                                         (array-dimension (tableau-matrix art-tab) 1)
                                         (max-problem-p main-tab) factor max-pivots n-pivots)))
                            (signal-outcome status)
                            (if full-tableau
                                (download-tableau main-handle main-tab main-flat main-basis)
                                (download-solution main-handle main-tab main-basis)))
                       (%tab-destroy main-handle)))
                (%tab-destroy art-handle)))))
          ;; single phase, src/simplex.lisp:453-461
          (if (and (> (device-count-of devices) 1)
                   (not (eq :overflowed
                            (solve-column-partitioned tableaus devices factor max-pivots
                                                      full-tableau n-pivots))))
              tableaus
              ;; one device.  With the light read-back (the default) only [non-basic columns | RHS]
              ;; is coerced and uploaded (mi355x_tab_create_compact); :full-tableau, or a basis that
              ;; is not a set of exact unit columns, takes the dense upload.
              (multiple-value-bind (compact-handle compact-basis)
                  (if full-tableau (values nil nil) (upload-tableau-compact tableaus device))
                (if compact-handle
                    (unwind-protect
                         (let ((status (solve-in-chunks
                                        (lambda (cap) (%tab-solve compact-handle (max-problem-p tableaus)
                                                                  factor cap n-pivots))
                                        (1+ (tableau-constraint-count tableaus))
                                        (1+ (tableau-var-count tableaus)) max-pivots n-pivots)))
                           (signal-outcome status)
                           (download-solution compact-handle tableaus compact-basis))
                      (%tab-destroy compact-handle))
                    (multiple-value-bind (handle flat basis) (upload-tableau tableaus device)
                      (unwind-protect
                           (let ((status (solve-in-chunks
                                          (lambda (cap) (%tab-solve handle (max-problem-p tableaus) factor
                                                                    cap n-pivots))
                                          (1+ (tableau-constraint-count tableaus))
                                          (1+ (tableau-var-count tableaus)) max-pivots n-pivots)))
                             (signal-outcome status)
                             (if full-tableau
                                 (download-tableau handle tableaus flat basis)
                                 (download-solution handle tableaus basis)))
                        (%tab-destroy handle))))))))))

;;; ------------------------------------------------------------------ many problems at once
;;; The hook takes ONE problem per call (src/solver.lisp:53-56), so N calls of solve-problem solve
;;; N small LPs one after the other -- each alone cannot fill the GPU (a 512 x 256 LP occupies 8 of
;;; 256 CUs).  BASELINE config 4 is exactly that workload; the library solves such LPs side by
;;; side (mi355x_multibatch_*: every LP resident in the register files of a few CUs, all LPs of a
;;; sub-batch in one launch, one sub-batch per GPU, no communication).  This is its Lisp entry:
;;; a LIST of problems in, a list of solved tableaus out, each answering the four solution-*
;;; generics exactly as the tableau (solve-problem p) returns.
(defun outcome-condition (status)
  "The condition SIGNAL-OUTCOME would signal for STATUS, as an object (NIL for MI_OPTIMAL)."
  (handler-case (progn (signal-outcome status) nil)
    (error (c) c)))

(defun pack-tableaus (tableaus)
  "Same-shape tableaus -> (values flat bases rows cols): the matrices coerced to double-float one
after the other, the bases as (signed-byte 64)."
  (let* ((n (length tableaus))
         (first-matrix (tableau-matrix (first tableaus)))
         (rows (array-dimension first-matrix 0))
         (cols (array-dimension first-matrix 1))
         (m (1- rows))
         (flat (make-array (* n rows cols) :element-type 'double-float))
         (bases (make-array (max 1 (* n m)) :element-type '(signed-byte 64) :initial-element 0)))
    (loop for tab in tableaus for k from 0
          do (let ((matrix (tableau-matrix tab))
                   (basis (tableau-basis-columns tab))
                   (base (* k rows cols)))
               (dotimes (r rows)
                 (dotimes (c cols)
                   (setf (aref flat (+ base (* r cols) c))
                         (coerce (aref matrix r c) 'double-float))))
               (dotimes (i m)
                 (setf (aref bases (+ (* k m) i)) (aref basis i)))))
    (values flat bases rows cols)))

(defun multibatch-create (tableaus devices)
  "mi355x_multibatch_create over DEVICES (a count or a list of ids): LP k lives in sub-batch
k / ceil(n / devices)."
  (multiple-value-bind (flat bases rows cols) (pack-tableaus tableaus)
    (let ((n-dev (device-count-of devices)))
      (cffi:with-foreign-objects ((out :pointer) (ids :int (max n-dev 1)))
        (when (listp devices)
          (loop for d in devices for i from 0 do (setf (cffi:mem-aref ids :int i) d)))
        (cffi:with-pointer-to-vector-data (pm flat)
          (cffi:with-pointer-to-vector-data (pb bases)
            (check (with-foreign-fp-mode
                     (%multibatch-create out (length tableaus) rows cols pm pb n-dev
                                         (if (listp devices) ids (cffi:null-pointer)))))))
        (cffi:mem-ref out :pointer)))))

(defun multibatch-read-back (handle k tab full-tableau)
  "Member K of a batch into TAB's own arrays: every entry, or (default) the objective row, the RHS
column and the basis -- all that tableau-variable & co. read (src/simplex.lisp:74-120)."
  (let* ((matrix (tableau-matrix tab))
         (rows (array-dimension matrix 0))
         (cols (array-dimension matrix 1))
         (last-row (make-array cols :element-type 'double-float))
         (last-col (make-array rows :element-type 'double-float))
         (flat (when full-tableau (make-array (* rows cols) :element-type 'double-float)))
         (basis (make-array (max 1 (1- rows)) :element-type '(signed-byte 64) :initial-element 0))
         (basis-dst (tableau-basis-columns tab)))
    (cffi:with-pointer-to-vector-data (pr last-row)
      (cffi:with-pointer-to-vector-data (pc last-col)
        (cffi:with-pointer-to-vector-data (pb basis)
          (if full-tableau
              (cffi:with-pointer-to-vector-data (pm flat)
                (check (%multibatch-download handle k pm pb pr pc)))
              (check (%multibatch-download handle k (cffi:null-pointer) pb pr pc))))))
    (if full-tableau
        (vectors->tableau tab flat basis)
        (progn
          (dotimes (r rows) (setf (aref matrix r (1- cols)) (aref last-col r)))
          (dotimes (c cols) (setf (aref matrix (1- rows) c) (aref last-row c)))
          (dotimes (i (length basis-dst)) (setf (aref basis-dst i) (aref basis i)))
          tab))))

(defun multibatch-solve-in-chunks (handle is-max factor rows cols max-pivots n status pivots)
  "mi355x_multibatch_solve in bounded foreign calls, as for a single tableau: members a chunk left
at MI_MAX_PIVOTS carry on in the next call (finished ones re-price as optimal at once).  STATUS
(int32 x N) holds the members' statuses afterwards.  Returns the pivot budget used (chunk caps)."
  (let ((chunk (chunk-pivots rows cols))
        (done 0))
    (loop
      (let ((cap (if (plusp max-pivots) (min chunk (- max-pivots done)) chunk)))
        (check (with-foreign-fp-mode
                 (%multibatch-solve handle is-max factor cap status pivots)))
        (incf done cap)
        (when (or (and (plusp max-pivots) (>= done max-pivots))
                  (loop for k below n
                        never (= (cffi:mem-aref status :int32 k) +mi-max-pivots+)))
          (return done))))))

(defun solve-two-phase-batch (pairs devices factor max-pivots full-tableau)
  "PAIRS: lists (art-tableau main-tableau) of ONE shape each and one sense (build-tableau's results
for two-phase problems, src/simplex.lisp:326-328).  Phase 1 on the batch of artificial tableaux in
bounded calls, the step between the phases per member on the devices
(mi355x_multibatch_two_phase_handover: feasibility test, drive-out pivots, hand-over;
src/simplex.lisp:405-451), phase 2 on the batch of main tableaux in bounded calls.  MAX-PIVOTS (0 =
none) caps each phase.  Returns a list parallel to PAIRS: the solved main tableau or a condition
object."
  (let* ((n (length pairs))
         (art-matrix (tableau-matrix (first (first pairs))))
         (rows (array-dimension art-matrix 0))
         (cols (array-dimension art-matrix 1))
         (art-handle (multibatch-create (mapcar #'first pairs) devices))
         (main-handle nil))
    (unwind-protect
         (progn
           (setf main-handle (multibatch-create (mapcar #'second pairs) devices))
           (cffi:with-foreign-objects ((status1 :int32 n) (between :int32 n) (status2 :int32 n)
                                       (pivots :int64 n))
             (multibatch-solve-in-chunks art-handle 0 factor rows cols max-pivots n status1 pivots)
             (check (with-foreign-fp-mode
                      (%multibatch-two-phase-handover art-handle main-handle factor status1 between
                                                      (cffi:null-pointer))))
             (multibatch-solve-in-chunks main-handle (max-problem-p (second (first pairs))) factor
                                         rows cols max-pivots n status2 pivots)
             (loop for (nil main-tab) in pairs for k from 0
                   collect (let ((st (if (= (cffi:mem-aref between :int32 k) +mi-optimal+)
                                         (cffi:mem-aref status2 :int32 k)
                                         (cffi:mem-aref between :int32 k))))
                             (or (outcome-condition st)
                                 (multibatch-read-back main-handle k main-tab full-tableau))))))
      (when main-handle (%multibatch-destroy main-handle))
      (%multibatch-destroy art-handle))))

(defun solve-same-shape-batch (tableaus devices factor max-pivots full-tableau)
  "TABLEAUS: single-phase tableaus of ONE shape and ONE sense.  Packs them into one multi-device
batch, solves them side by side and writes every member's results back into its own arrays.
Returns a list parallel to TABLEAUS: the tableau, or a condition object for a member without a
solution (unbounded-problem-error ...)."
  (let* ((n (length tableaus))
         (matrix (tableau-matrix (first tableaus)))
         (rows (array-dimension matrix 0))
         (cols (array-dimension matrix 1))
         (handle (multibatch-create tableaus devices)))
    (unwind-protect
         (cffi:with-foreign-objects ((status :int32 n) (pivots :int64 n))
           (multibatch-solve-in-chunks handle (max-problem-p (first tableaus)) factor rows cols
                                       max-pivots n status pivots)
           (loop for tab in tableaus for k from 0
                 collect (or (outcome-condition (cffi:mem-aref status :int32 k))
                             (multibatch-read-back handle k tab full-tableau))))
      (%multibatch-destroy handle))))

(defun mi355x-solve-problems (problems &rest args
                              &key (fp-tolerance 1024) (device 0) (devices 1) (max-pivots 0)
                                full-tableau (errorp t) native
                              &allow-other-keys)
  "Solves a LIST of problems and returns the list of their solved tableaus, in order -- what
  (mapcar #'solve-problem problems) returns, with the independent LPs running side by side on
the GPU(s) instead of one after the other.
  * Single-phase problems (every row a <= row after build-tableau's sign normalisation,
    src/simplex.lisp:243-263) are grouped by tableau shape and sense; a group of two or more is one
    multi-device batch (mi355x_multibatch_*: DEVICES sub-batches, one per GPU, no communication).
  * Two-phase problems (build-tableau returned (art main), src/simplex.lisp:326-328) are grouped by
    the shapes of their two tableaux; a group of two or more is a pair of batches -- phase 1, the
    per-member feasibility test, drive-out pivots and hand-over (src/simplex.lisp:402-452) and phase 2
    on the devices, each phase in bounded foreign calls (mi355x_multibatch_solve with a cap,
    mi355x_multibatch_two_phase_handover between them).  Problems alone in their group go through
    MI355X-SIMPLEX-SOLVER one by one.
  * A member without a solution does not abort the others: with :ERRORP NIL its place in the
    result holds the condition object (unbounded-problem-error, infeasible-problem-error,
    unsupported-constraint-error ...); with :ERRORP T (default, what mapcar of solve-problem
    would do) the first such condition is signalled after every member has been attempted.
  * :NATIVE :MANY hands the whole list to the library instead (mi355x_simplex_solver_many_*: the
    problems marshalled as they are, tableaux assembled, grouped and batched in C++): the result list
    then holds MI355X-SOLUTION objects -- the way to solve BASELINE config 4's 1 024 LPs from Lisp
    without 1 024 boxed tableaux.  (:NATIVE NIL, the default, keeps every member a `tableau`.)
  * The other keywords are MI355X-SIMPLEX-SOLVER's, applied to every member: :FP-TOLERANCE (the
    tolerance factor, src/simplex.lisp:506-511), :DEVICE (the GPU of members solved alone), :DEVICES
    (a count or a list of device ids: the sub-batches' GPUs), :MAX-PIVOTS (a cap per member; 0 = none,
    as the reference), :FULL-TABLEAU (every entry of every solved tableau written back).
Every returned solution object's results are bit-identical to the single-problem path's."
  (declare (ignore args))
  (when (and (eq native :many) (not full-tableau) problems)
    ;; the whole list behind ONE job of the library: no build-tableau, no boxed matrices; the members
    ;; come back as MI355X-SOLUTION objects
    (let ((results (solve-problems-natively problems (coerce fp-tolerance 'double-float)
                                            (if (and (integerp devices) (<= devices 1)) (list device) devices)
                                            max-pivots)))
      (when errorp
        (let ((failed (find-if (lambda (r) (typep r 'condition)) results)))
          (when failed (error failed))))
      (return-from mi355x-solve-problems results)))
  (let* ((n (length problems))
         (results (make-array n :initial-element nil))
         (groups (make-hash-table :test #'equal))
         (groups2 (make-hash-table :test #'equal))          ; two-phase members, by the shapes of (art main)
         (factor (coerce fp-tolerance 'double-float)))
    (flet ((solve-alone (k problem)
             (setf (aref results k)
                   ;; (:native NIL, the default here, keeps the result list homogeneous -- every member
                   ;; a `tableau`; :native :auto lets a member solved alone take the native route)
                   (handler-case (mi355x-simplex-solver problem :fp-tolerance fp-tolerance
                                                                :device device :max-pivots max-pivots
                                                                :full-tableau full-tableau
                                                                :native native)
                     (error (c) c)))))
      ;; build-tableau for every member; two-phase members and integer problems leave the batch
      (loop for problem in problems for k from 0
            do (if (problem-integer-vars problem)
                   (solve-alone k problem)             ; -> unsupported-constraint-error
                   (let ((tableaus (handler-case
                                       (build-tableau problem problem :fp-tolerance-factor fp-tolerance)
                                     (error (c) c))))
                     (cond
                       ((typep tableaus 'condition) (setf (aref results k) tableaus))
                       ((listp tableaus)
                        (let ((art (tableau-matrix (first tableaus)))
                              (main (tableau-matrix (second tableaus))))
                          (push (cons k tableaus)
                                (gethash (list (array-dimension art 0) (array-dimension art 1)
                                               (array-dimension main 1) (max-problem-p (second tableaus)))
                                         groups2))))
                       ((not (unit-basis-p tableaus)) (solve-alone k problem))
                       (t
                        (let ((matrix (tableau-matrix tableaus)))
                          (push (cons k tableaus)
                                (gethash (list (array-dimension matrix 0) (array-dimension matrix 1)
                                               (max-problem-p tableaus))
                                         groups))))))))
      (maphash
       (lambda (shape members)
         (declare (ignore shape))
         (setf members (reverse members))
         (if (rest members)
             (loop for (k . nil) in members
                   for outcome in (solve-same-shape-batch (mapcar #'cdr members) devices factor
                                                          max-pivots full-tableau)
                   do (setf (aref results k) outcome))
             (solve-alone (car (first members)) (nth (car (first members)) problems))))
       groups)
      (maphash
       (lambda (shape members)
         (declare (ignore shape))
         (setf members (reverse members))
         (if (rest members)
             (loop for (k . nil) in members
                   for outcome in (solve-two-phase-batch (mapcar #'cdr members) devices factor
                                                         max-pivots full-tableau)
                   do (setf (aref results k) outcome))
             (solve-alone (car (first members)) (nth (car (first members)) problems))))
       groups2))
    (when errorp
      (let ((failed (find-if (lambda (r) (typep r 'condition)) results)))
        (when failed (error failed))))
    (coerce results 'list)))
